cd $GRAFT_REPO_ROOT
for pp in 1 0; do LMH_OPT_CONV_PP=$pp python bench.py --no-cpu-baseline --no-other-configs --no-rocprof --steps 10 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pp=$pp', d['ms_per_step'])
for k,v in d['roofline']['all_conv_kernels'].items(): print('   %-40s x%5.1f %7.1f TF/s %7.3f ms'%(k,v['launches_per_step'],v['tflops'],v['ms_per_step']))
" | head -12; done
