"""SURVEY §8(f)1 (scoring fused into the chunk's attention) is NOT built; this pins the evidence for that decision
(tools/f1_logit_overlap.py): with pos_embed_reforge - every shipped config - the score contracts un-rotated
q~ k~ while the attention contracts rotated q k, and the two logit matrices are different matrices."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_unrotated_and_rotated_logits_share_nothing_with_reforge_on():
    import f1_logit_overlap as f1

    r = f1.logit_overlap(grids=8, gh=14, gw=14, seed=0)   # 0.91 / 1.25 / 2.89; a whole 32-frame chunk is further apart
    assert r["corr"] < 0.95, r                       # not the same matrix: no Q K^T to share
    assert r["rms_diff"] > 0.25 * r["logit_std"], r   # and not a small perturbation of it either
    assert r["kept_set_overlap_at_ratio_0.25"] < 0.9, r   # scoring on the attention's logits would keep other tokens


def test_fusion_ceiling_is_three_percent_of_attention_with_reforge_off():
    import f1_logit_overlap as f1

    for kw in ({}, {"L": 2304, "keep": 576}):
        f = f1.flop_shares(**kw)
        assert abs(f["score_over_attention_first_chunk"] - 2.0) < 1e-12
        assert 0.11 < f["score_over_attention_video"] < 0.13
        assert f["ceiling_saving_vs_attention_reforge_off"] < 0.03 and f["ceiling_saving_reforge_on"] == 0.0
