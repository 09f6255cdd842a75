"""Unit-level pins of the oracle's value semantics: the reference's golden cases for integer handling
(test/filestore/json/default/cases/case_integer.json) plus the literal rules of value/*.go,
expression/arith_*.go, comp_*.go and logic_*.go that the end-to-end cases do not reach."""
import numpy as np
import pytest

from oracle import n1o

T = n1o


def table(**cols):
    """cols: name -> list of python values (None = NULL, 'MISSING' = MISSING, bool, int, float, str)."""
    strings, index = [], {}
    out = []
    for name, vals in cols.items():
        tags = np.zeros(len(vals), np.uint8)
        pay = np.zeros(len(vals), np.uint64)
        for i, v in enumerate(vals):
            if isinstance(v, str) and v == "MISSING":
                tags[i] = T.T_MISSING
            elif v is None:
                tags[i] = T.T_NULL
            elif v is True or v is False:
                tags[i] = T.T_TRUE if v else T.T_FALSE
            elif isinstance(v, int):
                tags[i], pay[i] = T.T_INT, np.int64(v).view(np.uint64)
            elif isinstance(v, float):
                tags[i], pay[i] = T.T_FLOAT, np.float64(v).view(np.uint64)
            else:
                b = v.encode()
                if b not in index:
                    index[b] = len(strings)
                    strings.append(b)
                tags[i], pay[i] = T.T_STRING, index[b]
        out.append(n1o.Column("(`d`.`%s`)" % name, n1o.COL_TAGGED64, tags=tags, payload=pay))
    return n1o.Table(out, strings)


ONE_ROW = table(x=[0])


def ev(expr, t=ONE_ROW):
    return n1o.eval_expr(t, expr)


def test_case_integer_golden_arithmetic():
    # SELECT 9007199254740993 + 0, * 1, -x, x - 0  -> all stay exact int64 (case_integer.json case 2)
    big = 9007199254740993
    assert ev("(%d + 0)" % big)[0] == (T.T_INT, big)
    assert ev("(%d * 1)" % big)[0] == (T.T_INT, big)
    assert ev("(-%d)" % big)[0] == (T.T_INT, -big)
    assert ev("(%d - 0)" % big)[0] == (T.T_INT, big)
    # 9007199254740993.0 is a float64 literal: folds to the int64 its float value equals (case 1)
    assert ev("9007199254740993.0")[0] == (T.T_INT, 9007199254740992)
    # IDIV(5,2)=2, DIV 5/2=2.5, IMOD(5,2)=1, IDIV(5,0)=NULL, IMOD(5,0)=NULL (case 3)
    assert ev("idiv(5, 2)")[0] == (T.T_INT, 2)
    assert ev("(5 / 2)")[0] == (T.T_FLOAT, 2.5)
    assert ev("imod(5, 2)")[0] == (T.T_INT, 1)
    assert ev("idiv(5, 0)")[0][0] == T.T_NULL
    assert ev("imod(5, 0)")[0][0] == T.T_NULL


def test_int_add_same_sign_rule_and_overflow():
    # value/integer.go:266-277: int64 only for same-sign operands without overflow
    assert ev("(5 + 3)")[0] == (T.T_INT, 8)
    assert ev("(5 + -3)")[0] == (T.T_FLOAT, 2.0)       # mixed signs -> float64
    assert ev("(5 - 3)")[0] == (T.T_FLOAT, 2.0)        # Sub = Add(-n)
    assert ev("(9223372036854775807 + 1)")[0] == (T.T_FLOAT, 9.223372036854775807e18)
    assert ev("(-5 + -3)")[0] == (T.T_FLOAT, -8.0)     # Add starts from int 0: 0 + (-5) is already mixed-sign
    assert ev("(4 / 2)")[0] == (T.T_INT, 2)            # Div folds integral results (arith_div.go:58-59)
    assert ev("(7 % 4)")[0] == (T.T_INT, 3)
    assert ev("(3037000500 * 3037000500)")[0][0] == T.T_FLOAT  # overflow -> float


def test_missing_null_propagation():
    t = table(a=[1, None, "MISSING", "s"], b=[2, 2, 2, 2])
    assert [v[0] for v in ev("((`d`.`a`) + (`d`.`b`))", t)] == [T.T_INT, T.T_NULL, T.T_MISSING, T.T_NULL]
    assert [v[0] for v in ev("((`d`.`a`) < (`d`.`b`))", t)] == [T.T_TRUE, T.T_NULL, T.T_MISSING, T.T_FALSE]  # string > number
    assert [v[0] for v in ev("((`d`.`a`) = (`d`.`b`))", t)] == [T.T_FALSE, T.T_NULL, T.T_MISSING, T.T_FALSE]
    assert [v[0] for v in ev("((`d`.`a`) is valued)", t)] == [T.T_TRUE, T.T_FALSE, T.T_FALSE, T.T_TRUE]
    assert [v[0] for v in ev("((`d`.`a`) is null)", t)] == [T.T_FALSE, T.T_TRUE, T.T_MISSING, T.T_FALSE]
    assert [v[0] for v in ev("((`d`.`a`) between 0 and 5)", t)] == [T.T_TRUE, T.T_NULL, T.T_MISSING, T.T_FALSE]


def test_four_valued_logic_tables():
    vals = {"T": True, "F": False, "N": None, "M": "MISSING"}
    names = list(vals)
    a = [vals[x] for x in names for _ in names]
    b = [vals[y] for _ in names for y in names]
    t = table(a=a, b=b)
    tag = {T.T_TRUE: "T", T.T_FALSE: "F", T.T_NULL: "N", T.T_MISSING: "M"}
    got_and = "".join(tag[v[0]] for v in ev("((`d`.`a`) and (`d`.`b`))", t))
    got_or = "".join(tag[v[0]] for v in ev("((`d`.`a`) or (`d`.`b`))", t))
    # And.Apply: FALSE if any false, else MISSING > NULL > TRUE.  Or.Apply: TRUE if any true, else NULL > MISSING > FALSE
    assert got_and == "TFNM" "FFFF" "NFNM" "MFMM"
    assert got_or == "TTTT" "TFNM" "TNNN" "TMNM"
    assert "".join(tag[v[0]] for v in ev("(not (`d`.`a`))", table(a=[True, False, None, "MISSING"]))) == "FTNM"


def test_collation_across_types_and_min_max():
    t = table(g=[1] * 7, v=[3, 2.5, "abc", True, None, "MISSING", False])
    r = n1o.run(t, None, ["(`d`.`g`)"], ["max((`d`.`v`))", "min((`d`.`v`))", "count((`d`.`v`))", "countn((`d`.`v`))"])
    mx, mn, cnt, cntn = r.aggs[0]
    assert mx == (T.T_STRING, b"abc") and mn == (T.T_FALSE, None)  # BOOLEAN < NUMBER < STRING
    assert cnt == (T.T_INT, 5) and cntn == (T.T_INT, 2)


def test_count_scan_config1(tmp_path):
    """BASELINE config 1: SELECT COUNT(*) on datastore/file = number of directory entries
    (datastore/file/file.go:296-302; CountScan, execution/scan_count.go:55)."""
    n = 2000
    for i in range(n):
        (tmp_path / ("d%d.json" % i)).write_text('{"id":"d%d","cat":"cat_%d","price":%d}' % (i, i % 7, i % 100))
    assert n1o.count_scan(str(tmp_path)) == n


@pytest.mark.slow
def test_count_scan_config1_100k(tmp_path):
    n = 100_000
    for i in range(n):
        with open(tmp_path / ("d%d.json" % i), "w") as f:
            f.write('{"id":"d%d"}' % i)
    assert n1o.count_scan(str(tmp_path)) == n
