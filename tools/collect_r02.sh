#!/bin/bash
# Round-2 measurement batch on the GPU box: everything that DESIGN.md section 7 quotes.  Writes gpurun_out/r02/.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02
mkdir -p $OUT
python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err
python bench.py --no-cpu-baseline --phase-profile > /dev/null 2> $OUT/phase_profile.txt
for cfg in "1 4" "32 8" "128 8"; do set -- $cfg
  python bench.py --no-cpu-baseline --spectra $1 --chains $2 --steps 5 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('units %d: %.4g evals/s, %.2f us per round' % (d['config']['units_per_gpu'], d['value'], d['ms_per_step']))"
done > $OUT/few_chain_points.txt 2>&1
python bench.py --no-cpu-baseline --spectra 1 --chains 4 --steps 5 --warmup 5 --phase-profile 2>&1 | grep SOLO > $OUT/solo_phase_profile.txt
python tools/config3_run.py > $OUT/config3.txt 2>&1
python tools/bench_config5.py 4096 --phase-profile > $OUT/config5.txt 2>&1
python tools/bench_config5.py 4096 --series-outliers >> $OUT/config5.txt 2>&1
python tools/map_timing.py > $OUT/map_timing.txt 2>&1
python tools/bench_logp.py > $OUT/logp_micro.txt 2>&1
bash tools/profile_sq.sh r02 > $OUT/sq.log 2>&1
cp gpurun_out/sq_r02/summary.txt $OUT/sq_counters_nuts_kernel.txt 2>/dev/null
ls -la $OUT
