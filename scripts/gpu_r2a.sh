# Round 2, GPU call A: all GPU tests, smoke, the default bench line, rocprof summaries (production + serial schedule),
# VGG conditioning study, ds_read_b64_tr_b16 probe.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/r2a_tests.log 2>&1; tail -25 gpurun_out/r2a_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err; tail -c 2500 gpurun_out/r2a_bench.json; tail -3 gpurun_out/r2a_bench.err
python scripts/check_vgg_conditioning.py > gpurun_out/r2a_vgg_cond.log 2>&1; tail -12 gpurun_out/r2a_vgg_cond.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/probes/tr_b16_probe.hip -o /tmp/tr_probe > /dev/null 2>&1 && /tmp/tr_probe > gpurun_out/r2a_tr_probe.log 2>&1; head -70 gpurun_out/r2a_tr_probe.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2a -o r02 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_r2a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2a_serial -o r02 -- python $R/bench.py --serial --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_r2a_serial.log 2>&1
cd $R; ls gpurun_out/prof_r2a gpurun_out/prof_r2a_serial
