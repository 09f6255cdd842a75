// n1k_json.h — host side: raw JSON documents -> the plan's leaf columns.
//
// What the reference does per row and per referenced field while the operators run — Field.Apply on a parsedValue
// (expression/nav_field.go:134-160, value/parsed.go:159-207: go_json.FirstFind of the field in the raw bytes, then
// value.NewValue typing, value/value.go:367-430) — done once per batch, for the leaf paths only.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace n1k {

struct JsonStep {
    std::string name;     // field navigation (expression/nav_field.go:134-160)
    long long index = 0;  // element navigation with a constant integer index (expression/nav_element.go:49-65; negative
    bool is_index = false;  // counts from the end, value/array.go:204-214)
};
struct JsonPath {
    std::vector<JsonStep> names;  // steps below the keyspace alias: ((`default`.`a`)[1]) -> {a, [1]}
};

// (`alias`.`f1`...`fn`), ((`alias`.`f`)[i]) / nested parenthesised forms of expression.Stringer -> steps; false when the
// path holds anything but field names and constant integer indices
bool parse_leaf_path(const std::string& text, JsonPath& out);

struct JsonColumns {
    std::vector<std::vector<uint8_t>> tags;      // per path, per document: n1k_tag
    std::vector<std::vector<uint64_t>> payload;  // INT: int64, FLOAT: bits, STRING/ARRAY/OBJECT: index into `strings`
    std::vector<std::string> strings;            // distinct string bytes / canonical array and object texts, per call
};

// Parses documents [first, last) of the batch.  Returns -1 on success, else the index of the first malformed document.
long long extract_json_range(const std::vector<JsonPath>& paths, const uint64_t* offsets, const char* bytes, uint64_t first,
                             uint64_t last, JsonColumns& out, std::string& err);

// value.MarshalJSON pieces shared with the engine (ARRAY_AGG builds canonical array text): a float as
// strconv.FormatFloat(f, 'f', -1, 64) prints it (value/float.go:31-48), a string as a JSON string without HTML escaping
void format_float(double f, std::string& o);
void json_quote(const std::string& s, std::string& o);

}  // namespace n1k
