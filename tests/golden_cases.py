"""Loader / checker for tests/golden/reference_known_answers.json (the reference's own test data)."""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    with open(os.path.join(HERE, "golden", "reference_known_answers.json")) as f:
        return json.load(f)["cases"]


def case_id(c):
    return f'{c["op"]}@{c["ref"]}'


def input_of(c):
    if "x" in c:
        return np.asarray(c["x"], dtype=np.float64)
    s = c["x_sparse"]
    x = np.zeros(s["n"], dtype=np.float64)
    x[s["at"]:s["at"] + len(s["values"])] = s["values"]
    return x


def _check_value(name, got, spec, atol, rtol):
    if isinstance(spec, dict):  # bounds
        g = float(got)
        if "ge" in spec: assert g >= spec["ge"], f"{name} = {g} !>= {spec['ge']}"
        if "gt" in spec: assert g > spec["gt"], f"{name} = {g} !> {spec['gt']}"
        if "le" in spec: assert g <= spec["le"], f"{name} = {g} !<= {spec['le']}"
        if "lt" in spec: assert g < spec["lt"], f"{name} = {g} !< {spec['lt']}"
    elif isinstance(spec, list):
        got = np.asarray(got, dtype=np.float64)
        assert len(got) == len(spec), f"{name}: length {len(got)} != {len(spec)}"
        if all(not isinstance(s, dict) for s in spec):
            np.testing.assert_allclose(got, np.asarray(spec, np.float64), rtol=rtol, atol=atol, err_msg=name)
        else:
            for j, s in enumerate(spec):
                _check_value(f"{name}[{j}]", got[j], s, atol, rtol)
    else:
        g = float(got)
        assert math.isclose(g, float(spec), rel_tol=rtol, abs_tol=atol), f"{name} = {g} != {spec}"


def check(c, result):
    """result: scalar / 1-d array for single-output ops, dict of fields otherwise
    (ragged fields as 1-d arrays; `n_<field>` expectations are their lengths)."""
    atol, rtol = c.get("atol", 0.0), c.get("rtol", 0.0)
    exp = c["expect"]
    if isinstance(exp, dict) and not any(k in exp for k in ("ge", "gt", "le", "lt")):
        for k, spec in exp.items():
            if k.startswith("n_"):
                assert len(result[k[2:]]) == spec, f"len({k[2:]}) = {len(result[k[2:]])} != {spec}"
            else:
                _check_value(k, result[k], spec, atol, rtol)
        ns = {k: (np.asarray(v, np.float64) if not np.isscalar(v) else v) for k, v in result.items()}
        for rel in c.get("relations", []):
            assert eval(rel, {"abs": abs, "__builtins__": {}}, ns), f"relation failed: {rel}  with {ns}"
    else:
        _check_value(c["op"], result, exp, atol, rtol)
