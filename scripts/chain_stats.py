"""Median / mean duration of the proposal -> RCNN chain kernels over every step of a rocprofv3 kernel trace.
usage: python scripts/chain_stats.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

NAMES = ('k_rpn_decode', 'k_sort_local', 'k_nms_mask', 'k_nms_reduce', 'k_rcnn_target', 'k_roi_pool_mean_fwd', 'k_head_fwd',
         'k_skinny_fwd', 'k_softmax', 'k_rcnn_loss(', 'k_rcnn_loss_grad', 'k_act_bwd<false>', 'k_conv_bwd_data_gen',
         'k_roi_sample_table', 'k_roi_pool_bwd_slab', 'k_l2_reg', 'k_rpn_target_subsample', 'k_conv_stem7x7s2')
d = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r['Kernel_Name']
        for k in NAMES:
            if k in n:
                d[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0.0
for k in NAMES:
    v = sorted(d.get(k, []))
    if not v:
        continue
    per_step = len(v) / max(1, len(d['k_nms_reduce']))
    med = v[len(v) // 2]
    tot += med * per_step
    print('%-26s x%.0f/step  median %7.1f  mean %7.1f  min %7.1f  max %7.1f us' % (k, per_step, med, sum(v) / len(v), v[0], v[-1]))
print('sum of medians per step: %.1f us' % tot)
