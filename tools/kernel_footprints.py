"""Static footprint table of the library's kernels (round 4): registers, LDS and block size from the code-object metadata
of `hipcc -S --cuda-device-only` output, and whether a block of the kernel can be placed on a CU that already holds TWO
blocks of the step's dominant GEMM kernels (k_conv_fwd<128,128>: 256 threads, 188 VGPR+AGPR -> 192 allocated, 68 KB LDS).

Why it matters: what inflates a latency-bound kernel of the proposal -> RCNN chain inside the step is
waiting for a CU with room, not contention once it runs — k_rcnn_loss took 31 us alone and 97-185 us inside the step
while its blocks needed a CU with only one resident GEMM block.

    python tools/kernel_footprints.py file.s [...] [--all]

Dynamic LDS (extern __shared__) is not in the metadata; the known cases are listed in DYN_LDS below."""
import re
import subprocess
import sys

# dynamic shared memory at the train-step shapes (bytes per block), from the launch sites
DYN_LDS = {'k_roi_pool_mean_fwd<8>': 131072, 'k_roi_pool_mean_fwd<4>': 65536, 'k_roi_pool_bwd_slab<4, true>': 131072,
           'k_roi_pool_bwd_slab<8, true>': 262144, 'k_sort_local': 33800, 'k_sort_merge_local': 33800}
# threads per block the step launches with, where it is below the kernel's launch bound
LAUNCH_THREADS = {'k_nms_mask': 64, 'k_rcnn_loss': 256, 'k_loss_mean': 64}
STEP = ['k_conv_fwd<128, 128, false>', 'k_conv_fwd<128, 64, false>', 'k_conv_fwd<64, 64, true>', 'k_conv_fwd<64, 64, false>',
        'k_conv_bwd_data<128, 128, false>', 'k_conv_bwd_data<128, 64, false>', 'k_conv_bwd_data<64, 64, false>',
        'k_wgrad_1x1<64, 64, 4>', 'k_conv_bwd_weight<64, 64, false, true>', 'k_conv_bwd_weight<64, 64, false, false>',
        'k_conv_stem7x7s2<0>', 'k_maxpool3_fwd', 'k_wino4_input', 'k_wino4_output<false, false>', 'k_wino4_output<false, true>',
        'k_wino4_dy', 'k_wino4_dw', 'k_wino4_weight_batch', 'k_colsum_finish', 'k_tail_reduce', 'k_tail_finish',
        'k_rpn_decode', 'k_sort_local', 'k_sort_global_multi<4>', 'k_sort_merge_local', 'k_gather_topk', 'k_nms_mask',
        'k_nms_reduce_p<true>', 'k_gather_keep', 'k_rcnn_target<false>', 'k_roi_pool_mean_fwd<8>', 'k_head_fwd',
        'k_rcnn_loss_grad', 'k_rcnn_loss', 'k_skinny_bwd_data', 'k_skinny_fwd', 'k_conv_bwd_weight_gen<64, 64>',
        'k_conv_bwd_data_gen<64, 64>', 'k_roi_sample_table', 'k_roi_pool_bwd_slab<4, true>', 'k_rpn_target_rowmax',
        'k_rpn_target_labels', 'k_rpn_target_subsample', 'k_rpn_loss', 'k_rpn_loss_grad', 'k_loss_mean', 'k_l2_reg',
        'k_sgd_momentum', 'k_conv_hs<1, 64, 64, false, true>', 'k_conv_hs<1, 64, 64, true, true>', 'k_conv_hs<1, 128, 64, false, true>',
        'k_wgrad_hs_tr<1, 64, 64, false, 4>', 'k_wgrad_hs_tr<1, 128, 128, true, 4>', 'k_maxpool_fwd_hs<1, true>', 'k_half_weights<1>', 'k_cast_to_half<1>']


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    return [re.sub(r'^void ', '', o).split('(')[0] for o in out]


def kernels(path):
    cur, res = {}, []
    for ln in open(path):
        m = re.match(r'\s+-? ?\.(agpr_count|group_segment_fixed_size|max_flat_workgroup_size|name|vgpr_count|sgpr_count):\s+(\S+)', ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'agpr_count' and 'name' in cur:          # a new kernel's block starts with its agpr_count line
            res.append(cur)
            cur = {}
        cur[k] = v
    if 'name' in cur:
        res.append(cur)
    return [r for r in res if 'vgpr_count' in r]


def main():
    files = [a for a in sys.argv[1:] if not a.startswith('--')]
    show_all = '--all' in sys.argv
    rows = []
    for f in files:
        ks = kernels(f)
        for k, n in zip(ks, demangle([k['name'] for k in ks])):
            rows.append((n, int(k['vgpr_count']), int(k.get('agpr_count', 0)), int(k['group_segment_fixed_size']),
                         int(k['max_flat_workgroup_size'])))
    seen = {}
    for r in rows:
        seen.setdefault(r[0], r)
    names = sorted(seen) if show_all else [n for n in STEP if n in seen]
    print('| kernel | threads / block | VGPR (+AGPR) | allocated per wave | waves / SIMD of one block | LDS per block | '
          'fits beside two resident 128x128 GEMM blocks |')
    print('|---|---|---|---|---|---|---|')
    for n in names:
        _, v, a, lds, thr = seen[n]
        thr = LAUNCH_THREADS.get(n, thr)
        lds += DYN_LDS.get(n, 0)
        alloc = -(-max(v, 1) // 8) * 8              # .vgpr_count already includes the AGPRs on gfx90a+
        wps = -(-thr // 256)                        # waves of one block on each SIMD
        free_regs, free_lds = 512 - 2 * 192, 160 * 1024 - 2 * 69632
        fits = wps * alloc <= free_regs and lds <= free_lds
        why = []
        if wps * alloc > free_regs:
            why.append('%d x %d registers > %d' % (wps, alloc, free_regs))
        if lds > free_lds:
            why.append('%d KB LDS > %d KB' % (lds // 1024, free_lds // 1024))
        print('| `%s` | %d | %d (%d) | %d | %d | %s | %s |' % (n, thr, v, a, alloc, wps, ('%.1f KB' % (lds / 1024.0)) if lds else '0',
                                                             'yes' if fits else 'no: ' + ', '.join(why)))


if __name__ == '__main__':
    main()
