cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --no-cpu-baseline --no-roofline --no-alt > gpurun_out/tmp.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/tmp.json')); print('%-56s %7.1f img/s %7.3f ms' % ('$*', d['value'], d['ms_per_step']))"; }
run A=0
run LUMINOTH_AMD_WINOGRAD_MIN_CK=16384
run LUMINOTH_AMD_WINOGRAD_WGRAD_MIN_CK=65536
run LUMINOTH_AMD_WINOGRAD_MIN_CK=16384 LUMINOTH_AMD_WINOGRAD_WGRAD_MIN_CK=16384
run A=1
