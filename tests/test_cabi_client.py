"""The boundary driven from plain C (tests/cabi_client.c, built with gcc, libn1k.so through dlopen): the call sequence of
the cgo operator of INTEGRATION.md — row-at-a-time staging into C buffers, n1k_push_batch every B rows, two operator
copies on two OS threads, n1k_stop from another thread while a third copy is being pushed to, the copies' groups
merged through n1k_export_groups / n1k_merge_groups, n1k_finish — checked against the oracle.
Reference: execution/execution.go:26-64, execution/parallel.go:67-83, execution/base.go:313-338."""
import os
import struct
import subprocess

import numpy as np
import pytest

import parity_util as pu
import query_amd
from oracle import n1o
from query_amd import _ffi, plan
from query_amd.gpu_operator import GroupRows

HERE = os.path.dirname(os.path.abspath(__file__))


def D(*names):
    return plan.field_path("default", *names)


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cabi") / "cabi_client")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-o", exe,
                           os.path.join(HERE, "cabi_client.c"), "-ldl", "-lpthread"])
    return exe


def test_c_client_binds_every_call_it_needs(client):
    """No GPU: the library loads from C, the symbols resolve, create / binding calls / stop / destroy work."""
    out = subprocess.run([client, _ffi.LIB_PATH, "symbols"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "symbols ok" in out.stdout


def _write_table(path, t, paths):
    by = {c.name: c for c in t.columns}
    cols = [by[p] for p in paths]
    with open(path, "wb") as f:
        f.write(struct.pack("<QII", t.nrows, len(cols), len(t.dictionary)))
        f.write(np.array([c.kind for c in cols], dtype=np.uint32).tobytes())
        for c in cols:
            if c.kind == n1o.COL_DICT32:
                f.write(np.ascontiguousarray(c.codes, dtype=np.uint32).tobytes())
            else:
                f.write(np.ascontiguousarray(c.tags, dtype=np.uint8).tobytes())
                f.write(np.ascontiguousarray(c.payload, dtype=np.uint64).tobytes())
        for s in t.dictionary:
            f.write(struct.pack("<I", len(s)))
            f.write(bytes(s))


def _parse_value(tok):
    kind, body = tok[0], tok[1:]
    if kind == "M":
        return (n1o.T_MISSING, None)
    if kind == "N":
        return (n1o.T_NULL, None)
    if kind in "FT":
        return (n1o.T_TRUE if kind == "T" else n1o.T_FALSE, None)
    if kind == "I":
        return (n1o.T_INT, int(body))
    if kind == "D":
        return (n1o.T_FLOAT, float(body))
    return ({"S": n1o.T_STRING, "A": n1o.T_ARRAY, "O": n1o.T_OBJECT}[kind], bytes.fromhex(body))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["filter-sum-min-max", "two-keys-no-filter"])
def test_c_client_plays_the_cgo_operator(client, tmp_path, shape):
    if shape == "filter-sum-min-max":
        cond, keys = "(50 < %s)" % D("price"), [D("cat")]
        aggs = sorted(["avg(%s)" % D("price"), "count(*)", "max(%s)" % D("price"), "min(%s)" % D("price"), "sum(%s)" % D("price")])
        t = n1o.synth_table(300_000, k_cat=37)
    else:
        cond, keys = None, [D("cat"), D("region_id")]
        aggs = sorted(["count(%s)" % D("price"), "sum(%s)" % D("user_id"), "max(%s)" % D("cat")])
        t = n1o.synth_table(200_001, k_cat=11)
    pj = plan.filter_group_plan(cond, keys, aggs)
    op = query_amd.GpuFilterGroup(pj)
    paths = op.column_paths
    op.done()
    plan_path, data_path, out_path = str(tmp_path / "plan.json"), str(tmp_path / "data.bin"), str(tmp_path / "groups.txt")
    with open(plan_path, "w") as f:
        f.write(pj)
    _write_table(data_path, t, paths)
    run = subprocess.run([client, _ffi.LIB_PATH, "run", plan_path, data_path, out_path, "8192"], capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0, run.stderr + run.stdout
    lines = open(out_path).read().splitlines()
    head = lines[0].split()
    rows_in = [int(x) for x in head[head.index("rows_in") + 1].split("+")]
    assert sum(rows_in) == t.nrows and min(rows_in) >= t.nrows // 2  # both copies took their half, batch by batch
    assert int(head[head.index("stopped_after") + 1]) >= 1
    got_keys, got_aggs = [], []
    for ln in lines[1:]:
        k, a = ln.split("|")
        got_keys.append(tuple(_parse_value(x) for x in k.split()))
        got_aggs.append(tuple(_parse_value(x) for x in a.split()))
    got = GroupRows(len(keys), len(aggs), got_keys, got_aggs, [])
    ora = n1o.run(t, cond, keys, aggs, threads=2)
    pu.assert_same_groups(got, ora, aggs=aggs)
