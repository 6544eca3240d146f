"""Host side of the ragged trigger columns (CPU tensors): compaction returns EVERY trigger, overflow is re-run or
fails loudly (reference src/intersect_maximum.jl:49-56 pushes every crossing; VERDICT r1 missing #4)."""
import pytest
import torch

import legenddsp_jl_amd as ldsp
from legenddsp_jl_amd.extractors import compact_fields, resolve_overflow, TriggerOverflow

FIELDS = ("x", "x_high", "x_tot", "max")


def _group(counts, cap, true_len=None):
    """slabs as a kernel would leave them: trace i, trigger j, field f -> 1000 i + 10 j + f; only the first cap stored"""
    n = len(counts)
    g = {"count": torch.tensor(counts, dtype=torch.int32)}
    for f, name in enumerate(FIELDS):
        s = torch.full((n, cap), float("nan"))
        for i, c in enumerate(counts):
            for j in range(min(c, cap)):
                s[i, j] = 1000 * i + 10 * j + f
        g[name] = s
    return g


def test_compact_fields_plain():
    g = _group([2, 0, 3, 1], 4)
    vals, cnt = compact_fields(g, FIELDS)
    assert cnt.tolist() == [2, 0, 3, 1]
    assert vals[:, 0].tolist() == [0, 10, 2000, 2010, 2020, 3000]
    assert vals[:, 3].tolist() == [3, 13, 2003, 2013, 2023, 3003]


def test_overflow_fails_loudly_without_second_pass():
    g = _group([2, 7, 1], 4)      # trace 1 counted 7 triggers, the slab holds 4
    with pytest.raises(TriggerOverflow):
        compact_fields(g, FIELDS)


def test_overflow_second_pass_returns_every_trigger():
    counts = [2, 7, 1, 9]
    groups = {"a": _group(counts, 4), "b": _group([1, 1, 1, 1], 4)}
    calls = []

    def rerun(rows, cap):
        calls.append((rows.tolist(), cap))
        sub = [counts[i] for i in rows.tolist()]
        ga = _group(sub, cap)
        for f, name in enumerate(FIELDS):      # the re-run slabs carry the ORIGINAL trace numbers
            for k, i in enumerate(rows.tolist()):
                for j in range(sub[k]):
                    ga[name][k, j] = 1000 * i + 10 * j + f
        return {"a": ga, "b": _group([1] * len(sub), cap)}

    groups = resolve_overflow(groups, rerun)
    assert calls == [([1, 3], 16)]            # only the overflowing traces, capacity = next power of two >= 9
    vals, cnt = compact_fields(groups["a"], FIELDS)
    assert cnt.tolist() == counts and vals.shape == (sum(counts), 4)
    exp = [1000 * i + 10 * j for i, c in enumerate(counts) for j in range(c)]
    assert vals[:, 0].tolist() == exp
    vb, cb = compact_fields(groups["b"], FIELDS)     # the other group of the re-run traces stays consistent
    assert cb.tolist() == [1, 1, 1, 1] and vb.shape == (4, 4)


def test_no_overflow_no_second_pass():
    groups = {"a": _group([1, 4, 0], 4)}
    out = resolve_overflow(groups, lambda rows, cap: (_ for _ in ()).throw(AssertionError("must not run")))
    assert "overflow" not in out["a"]


def test_per_trace_argument_must_match_the_batch():
    """A per-trace threshold / time tensor of the wrong length would make the kernel read out of bounds: refused on the host."""
    from legenddsp_jl_amd.extractors import _per_trace
    assert _per_trace(2.5, 4, "cpu").tolist() == [2.5] * 4
    assert _per_trace(torch.tensor(1.5), 3, "cpu").tolist() == [1.5] * 3
    assert _per_trace(torch.arange(4.0), 4, "cpu").tolist() == [0.0, 1.0, 2.0, 3.0]
    with pytest.raises(ValueError):
        _per_trace(torch.arange(5.0), 4, "cpu")
    with pytest.raises(ValueError):
        _per_trace(torch.zeros(4, 1), 4, "cpu")
