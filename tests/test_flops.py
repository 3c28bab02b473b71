"""tools/flops.py re-derives SURVEY.md §8d's algorithmic MAC / FLOP totals from the layer shapes of the networks the
product builds (the figures `roofline.whole_step` in bench.py is priced with)."""
from tools import flops


def test_flop_totals_match_the_survey():
    tab = flops.check()
    r50 = tab['r50_1024']
    assert r50['feature_hw'] == (64, 64) and tab['r50_800x1333']['feature_hw'] == (50, 84)
    # what bench.py counts per step at B = 2 (profiles/r02_bench_line.json: whole_step.conv_flops = 834.35 GFLOP): the
    # train-step total minus the two input gradients nobody needs (block2/unit_1's conv1 and shortcut read the frozen
    # block1 output), 2 x (16384 px x 256 x (128 + 512)) MACs per image
    skipped_flops = 2 * (128 * 128 * 256 * (128 + 512))
    per_step = 2 * (2 * r50['train'] - skipped_flops)
    assert abs(per_step / 1e9 - 834.35) < 0.5, per_step / 1e9
