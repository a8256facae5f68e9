"""CPU: the reference's OWN test cases for the pinned half of the path, read from tests/golden/reference_cases.json --
which tests/golden/extract_reference_cases.py cuts out of /root/reference/pkg/lwepp/handlers/request_test.go (nothing
here is typed by hand; every case records the Go source lines it came from).  Each case is driven through the host
mirror of handleRequestHeaders / pickEndpoint (picker.py), i.e. through libeppk's eppk_subset_mask and
eppk_round_robin, AND through the oracle's restatements of the same two functions."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DOC = json.load(open(os.path.join(HERE, "golden", "reference_cases.json")))
CASES = {c["name"]: c for c in DOC["cases"]}
REF_FILE = "/root/reference/pkg/lwepp/handlers/request_test.go"


def _endpoints(pkg, case):
    return [pkg.picker.Endpoint(p["address"], p["port"]) for p in case["pods"]]


def _metadata(case):
    md = case["metadata"]
    ns, key = DOC["subset_namespace"], DOC["subset_key"]
    if md["kind"] == "absent":
        return None
    if md["kind"] == "other-key":
        return {ns: {"some-other-unrelated-key": "val"}}
    return {ns: {key: md["value"] if md["kind"] == "string" else list(md["values"])}}


def _headers(case):
    return [] if case.get("header") is None else [(pkg_header(), case["header"])]


def pkg_header():
    return "test-epp-endpoint-selection"


def _candidates_both(pkg, orc, endpoints, filt):
    """Candidates per libeppk and per the oracle (must agree)."""
    mask, n = pkg.picker.subset_mask(endpoints, filt)
    omask, on = orc.subset_mask([e.address for e in endpoints], [e.port for e in endpoints], filt)
    assert n == on and np.array_equal(mask, omask)
    return [e for i, e in enumerate(endpoints) if (int(mask[i >> 6]) >> (i & 63)) & 1]


def test_fixture_is_current_when_the_reference_is_present():
    """In the build container the committed JSON must be exactly what the extractor produces from the reference file."""
    if not os.path.exists(REF_FILE):
        pytest.skip("no /root/reference on this box: the committed fixture is all there is")
    assert hashlib.sha256(open(REF_FILE, "rb").read()).hexdigest() == DOC["source_sha256"]
    assert len(DOC["cases"]) >= 14


@pytest.mark.parametrize("name", [n for n, c in CASES.items() if "table" not in c])
def test_handle_request_headers_case(pkg, orc, name):
    case = CASES[name]
    endpoints = _endpoints(pkg, case)
    exp = case["expect"]
    if exp.get("headers_error"):
        with pytest.raises(pkg.picker.Unavailable):
            pkg.picker.handle_request_headers(endpoints, _headers(case), _metadata(case))
        assert orc.round_robin(C.c_uint64(0), 0) == -1
        return
    filt = pkg.picker.resolve_subset_filter(_headers(case), _metadata(case))
    cands = pkg.picker.handle_request_headers(endpoints, _headers(case), _metadata(case))
    assert [(e.address, e.port) for e in cands] == [(e.address, e.port) for e in _candidates_both(pkg, orc, endpoints, filt)]
    if "n_candidates" in exp:
        assert len(cands) == exp["n_candidates"]
    if "candidate_addresses" in exp:
        assert sorted(e.address for e in cands) == sorted(exp["candidate_addresses"])      # assert.ElementsMatch
    if "selected_ip" in exp:            # pickEndpoint with the reference's RoundRobinPicker over the candidates
        res = pkg.picker.RoundRobinPicker().Pick(None, cands)
        assert res.endpoint.rsplit(":", 1)[0] == exp["selected_ip"]
        ctr = C.c_uint64(0)
        assert cands[orc.round_robin(ctr, len(cands))].address == exp["selected_ip"]
    if "pick_relations" in exp:         # successive picks of ONE server (one counter)
        rr, ctr = pkg.picker.RoundRobinPicker(), C.c_uint64(0)
        got = [rr.Pick(None, cands).endpoint for _ in range(exp["n_picks"])]
        ogot = [cands[orc.round_robin(ctr, len(cands))].address for _ in range(exp["n_picks"])]
        for seq in (got, ogot):
            for rel in exp["pick_relations"]:
                a, b = seq[rel["a"] - 1], seq[rel["b"] - 1]
                assert (a != b) if rel["op"] == "ne" else (a == b)


@pytest.mark.parametrize("row", [(n, r["name"]) for n, c in CASES.items() if "table" in c for r in c["table"]])
def test_port_aware_table_row(pkg, orc, row):
    case = CASES[row[0]]
    r = next(x for x in case["table"] if x["name"] == row[1])
    endpoints = _endpoints(pkg, case)
    md = {DOC["subset_namespace"]: {DOC["subset_key"]: r["filter_value"]}}
    cands = pkg.picker.handle_request_headers(endpoints, [], md)
    want = sorted((p["address"], p["port"]) for p in r["expected_candidates"])
    assert sorted((e.address, e.port) for e in cands) == want
    filt = pkg.picker.resolve_subset_filter([], md)
    assert sorted((e.address, e.port) for e in _candidates_both(pkg, orc, endpoints, filt)) == want
