"""Pins the CPU oracle against outputs of the REAL reference (tests/golden/*.npz, made by
tests/golden/make_golden.py from /root/reference).  Runs on CPU."""
import numpy as np
import pytest
import torch

from common import (GOLD, LOGIT_TOL, MHA_KNOB_CASES, case_clip, check_masks, load_case, mha_knob_inputs, run_teacher_forced,
                    synth_model_state)
from oracle.aot_oracle import OracleEngine, OracleModel, mha_core


@pytest.mark.parametrize('case', ['c1_aott', 'c1b_aott_ragged', 'c2_r50_aotl', 'c3a_deaott', 'c3b_r50_deaotl', 'c3c_swinb_deaotl'])
def test_oracle_matches_reference_golden(case):
    c, g = load_case(case)
    _, _, sd = synth_model_state(c['model'])
    frames, mask, objs, out_size = case_clip(c)
    eng = OracleEngine(OracleModel(c['model'], sd))
    keep = set(c['keep_logits'])
    res = run_teacher_forced(eng, frames, mask, objs, out_size, g, keep)
    no = c['num_obj'] + 1
    for t, (l4, m) in res.items():
        check_masks(m, g, t, 'oracle')
        if l4 is not None:
            err = np.abs(l4[:no] - g['logits4_%d' % t]).max()
            assert err < 1e-4, 'frame %d logits4 err %g' % (t, err)        # far inside the 1e-3 bar
            assert err < LOGIT_TOL


def test_oracle_fp64_agrees_with_fp32():
    """The fp32 oracle's own rounding noise (vs an fp64 evaluation of the same weights) is ~1e-5: the
    1e-3 tolerance is a property of the algorithm, not of lucky summation order."""
    c, g = load_case('c1_aott')
    _, _, sd = synth_model_state(c['model'])
    frames, mask, objs, out_size = case_clip(c)
    outs = []
    for dt in (torch.float32, torch.float64):
        eng = OracleEngine(OracleModel(c['model'], sd, dtype=dt))
        with torch.no_grad():
            eng.add_reference_frame(frames[0].to(dt), mask.to(dt), objs)
            eng.match_propogate_one_frame(frames[1].to(dt))
            eng.decode_current_logits(out_size)
        outs.append(eng.pred_id_logits[0, :2].double())
    assert (outs[0] - outs[1]).abs().max().item() < 2e-4


@pytest.mark.parametrize('case', sorted(MHA_KNOB_CASES))
def test_oracle_attention_knobs_match_reference_module(case):
    """top_k / max_mem_len_ratio (attention.py:84-89,102-105): the oracle's mha_core against the REAL reference
    MultiheadAttention run with those constructor knobs (tests/golden/mha_knobs.npz)."""
    import os
    g = np.load(os.path.join(GOLD, 'mha_knobs.npz'))
    Q, K, V, H = mha_knob_inputs()
    sums = np.array([Q.double().sum().item(), K.double().sum().item(), V.double().sum().item()])
    assert np.allclose(sums, g['input_sums'], rtol=0, atol=1e-9), 'seeded inputs differ from the ones the golden was made with'
    out = mha_core(Q, K, V, H, **MHA_KNOB_CASES[case]).numpy()
    assert np.abs(out - g[case]).max() < 2e-6


def test_oracle_bounded_bank_keeps_first_and_most_recent():
    """long_term_mem_max (repo extension): the bank never exceeds the bound, always holds the reference frame's rows
    first, and an unbounded engine agrees with it until the bound is reached."""
    from utils.synth import synth_clip
    _, _, sd = synth_model_state('aott')
    frames, mask, objs, out_size = synth_clip(6, 6, (65, 81), (64, 80), 2)
    a = OracleEngine(OracleModel('aott', sd), long_term_mem_gap=1, long_term_mem_max=3)
    b = OracleEngine(OracleModel('aott', sd), long_term_mem_gap=1)
    with torch.no_grad():
        for e in (a, b):
            e.add_reference_frame(frames[0], mask, objs)
        first = a.long_term_memories[0][0].clone()
        for t in range(1, 6):
            outs = []
            for e in (a, b):
                e.match_propogate_one_frame(frames[t])
                outs.append(e.decode_current_logits(out_size))
                e.update_memory(mask)
            n = a.enc_hw
            assert a.long_term_memories[0][0].shape[0] == min(t + 1, 3) * n
            assert torch.equal(a.long_term_memories[0][0][:n], first)
            if t <= 3:      # frames 1..3 still see identical banks (bank of frame t is built from frames < t)
                assert torch.equal(outs[0], outs[1])
            else:
                assert not torch.equal(outs[0], outs[1])
            assert torch.equal(a.long_term_memories[0][0][-n:], b.long_term_memories[0][0][-n:])
