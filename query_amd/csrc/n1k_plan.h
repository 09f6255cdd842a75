// n1k_plan.h — host side: the reference's plan JSON and expression.Stringer text -> compiled plan.
//
// Mirrors, for the hot path only, what plan.(*Filter).UnmarshalJSON (plan/filter.go:55-71) and
// plan.(*InitialGroup).UnmarshalJSON (plan/group.go:72-103) do with expression/parser: parse the
// stringified expressions back into trees.  Anything outside the device subset is reported as
// "unsupported" so that the caller keeps the reference operators (N1K_UNSUPPORTED).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "n1k_types.h"

namespace n1k {

struct PlanError {
    bool unsupported = false;  // valid but outside the device subset
    std::string msg;
};

// expression tree (subset of expression/*.go)
enum class EK {
    Const, Path, Add, Sub, Mult, Div, Mod, Neg, IDiv, IMod, Eq, LT, LE, Between, And, Or, Not,
    IsNull, IsNotNull, IsMissing, IsNotMissing, IsValued, IsNotValued,
    Func  // numeric functions of one or two arguments (expression/func_num.go): fname
};

struct Expr {
    EK kind;
    std::vector<std::unique_ptr<Expr>> ch;
    // Const
    uint32_t ctag = T_NULL;
    uint64_t cpayload = 0;   // INT: int64, FLOAT: bits; STRING: filled with the dictionary code at bind time
    std::string cstr;        // STRING constant bytes
    // Path
    std::string text;        // exact stringer text, e.g. (`default`.`price`)
    // Func
    std::string fname;       // round | trunc | abs | ceil | floor | sign | sqrt
};

struct AggDef {
    uint32_t kind;  // AGG_*
    bool distinct = false;
    std::unique_ptr<Expr> operand;  // null for count(*)
    std::string text;               // agg.String(): key of the reference's "aggregates" attachment map
};

// one ORDER BY term over the groups (plan/order.go:51-79): its expression must be, text for text, one of the group
// keys or one of the aggregates (the Go glue resolves projection aliases before handing the plan over)
struct OrderTerm {
    std::string text;
    bool desc = false;
    int key_index = -1;  // >= 0: group key
    int agg_index = -1;  // >= 0: aggregate
    int proj_index = -1; // >= 0: a projection term (by its alias, or by its expression text)
};

// one result term of the InitialProject after the group operators (plan/project.go:73-110): its expression reads group
// keys and aggregates (algebra/aggregate.go:97-118 looks an aggregate up by its text in the "aggregates" attachment)
struct ProjectTerm {
    std::string text;  // expression.Stringer text
    std::string as;    // explicit alias ("" = none: the caller derives one as algebra.ResultTerm does)
};

struct ParsedPlan {
    bool has_filter = false;
    bool has_group = false;
    std::unique_ptr<Expr> condition;
    std::vector<std::unique_ptr<Expr>> keys;
    std::vector<std::string> key_texts;  // stringer text of every group key
    std::vector<AggDef> aggs;
    std::vector<std::string> paths;  // distinct leaf paths in column order
    int max_parallelism = 0;
    // Order / Offset / Limit over the final groups (execution/order.go, order_limit.go, offset.go, limit.go)
    bool has_order = false;
    std::vector<OrderTerm> order;
    int64_t limit = -1;  // < 0: none
    int64_t offset = 0;
    // HAVING: a Filter after the group operators (planner/build_select_sub.go:295), over group keys and aggregates
    bool has_having = false;
    std::string having_text;
    // InitialProject / FinalProject over the final groups (execution/project_initial.go:52-144, project_final.go:51-59)
    bool has_project = false;
    std::vector<ProjectTerm> project;
};

// Parse plan JSON (Sequence / Parallel / Filter / InitialGroup nodes, optionally followed by IntermediateGroup /
// FinalGroup — which the device operator subsumes — and Order / Offset / Limit over the groups).  Returns false and fills err on failure.
bool parse_plan_json(const char* json, size_t len, ParsedPlan& out, PlanError& err);

// Parse one stringified expression / aggregate.
std::unique_ptr<Expr> parse_expression(const std::string& s, PlanError& err);
bool parse_aggregate(const std::string& s, AggDef& out, PlanError& err);

// value.Collate of two arrays / objects given as canonical JSON text (numbers through float64); false when a text does
// not parse
bool json_text_collate(const std::string& a, const std::string& b, int& out);

}  // namespace n1k
