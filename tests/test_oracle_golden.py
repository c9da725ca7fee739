"""Pins the CPU oracle against outputs of the REAL reference (tests/golden/*.npz, made by
tests/golden/make_golden.py from /root/reference).  Runs on CPU."""
import numpy as np
import pytest
import torch

from common import LOGIT_TOL, case_clip, check_masks, load_case, run_teacher_forced, synth_model_state
from oracle.aot_oracle import OracleEngine, OracleModel


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
