#!/bin/bash
# Round-6 measurement batch on the GPU box: everything DESIGN.md section 7 (round 6) quotes.  Writes gpurun_out/r06/.
# Usage: tools/collect_r06.sh [part ...]   parts: bench trace sq wave soak map config5 waveom hmc fuzz   (default: bench trace sq wave soak config5)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06
mkdir -p $OUT
PARTS="${*:-bench trace sq ablate wave soak config5}"
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has bench; then
  python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err
  python bench.py --no-cpu-baseline --phase-profile > /dev/null 2> $OUT/phase_profile.txt
  python - > $OUT/bench_tables.txt <<'PY'
import json
d = json.load(open('gpurun_out/r06/bench_default_line.json'))
c = d['config']
print('headline: %.1f M evals/s, roofline.frac %.3f (frac_executed %.3f), %.2f ms per step' % (d['value'] / 1e6, d['roofline']['frac'], d['roofline']['frac_executed'], d['ms_per_step']))
print('mid occupancy (sampler kind 1: one chain per 512-thread workgroup, 3: one chain per wave, 0: sixteen chains per workgroup):')
for m in c['mid_occupancy']: print('  %5d units: %6.1f M evals/s (kind %d)' % (m['units'], m['evals_per_s'] / 1e6, m['sampler_kind']))
print('shapes (frequencies x basis functions; evaluator code; evals/s and dense-formulation roofline fraction at 4096 and 2048 units):')
for r in c['shapes']:
    print('  %3d x %3d  evaluator %d  4096 units: %6.1f M (kind %d, frac %.3f)   2048 units: %6.1f M (kind %d, frac %.3f)' % (
        r['nf'], r['K'], r['evaluator'], r['units_4096']['evals_per_s'] / 1e6, r['units_4096']['sampler_kind'], r['units_4096']['frac'],
        r['units_2048']['evals_per_s'] / 1e6, r['units_2048']['sampler_kind'], r['units_2048']['frac']))
print('few chains:', json.dumps(c['few_chains']))
print('cpu baseline:', json.dumps({k: v for k, v in d['cpu_baseline'].items() if k in ('value', 'cores', 'single_core', 'kind')}))
PY
fi
if has trace; then
  bash tools/profile_bench.sh r06 > $OUT/profile_bench.log 2>&1
  for f in bench_line.json kernel_stats.txt pmc_FETCH_SIZE.txt pmc_WRITE_SIZE.txt pmc_traffic.json; do cp gpurun_out/prof_r06/$f $OUT/ 2>/dev/null; done
fi
if has sq; then
  bash tools/profile_sq.sh r06 > $OUT/sq.log 2>&1
  cp gpurun_out/sq_r06/summary.txt $OUT/sq_counters_nuts_kernel.txt 2>/dev/null
fi
if has ablate; then
  # dynamic instruction counts of the headline kernel with one evaluator phase skipped at a time (BDRT_DEBUG_SKIP: 4 prior chain, 8 likelihood,
  # 2 backward GEMM, 15 the whole evaluation; results wrong on purpose): the per-phase mix of profiles/r06/isa_mix_nuts_kernel.txt
  export BDRT_BENCH_NO_TABLES=1
  SET="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA"
  rm -f $OUT/ablate_phases.txt
  for SK in 0 4 8 2 15; do
    BDRT_DEBUG_SKIP=$SK timeout 300 rocprofv3 --pmc $SET -d $OUT/ab$SK -o bench -- python3 bench.py --gpus 1 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/ab$SK.log 2>&1
    DB=$(find $OUT/ab$SK -name '*.db' | head -1)
    echo "== BDRT_DEBUG_SKIP=$SK" >> $OUT/ablate_phases.txt
    grep '"metric"' $OUT/ab$SK.log | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   %.1f M evals/s, %.3f ms per 1000 rounds' % (d['value']/1e6, d['ms_per_step']))" >> $OUT/ablate_phases.txt
    [ -n "$DB" ] && python3 - "$DB" >> $OUT/ablate_phases.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%nuts_kernel%' group by counter_name order by 1").fetchall()
for r in rows: print('   %-28s dispatches %3d  avg/dispatch %.6g  per wave and round %.1f' % (r + (r[2] / 2048 / 1000,)))
PY
    rm -rf $OUT/ab$SK $OUT/ab$SK.log
  done
  unset BDRT_BENCH_NO_TABLES
fi
if has wave; then
  python tools/wave_sweep.py 4 256 512 768 1024 1536 2048 2560 3072 4096 2>&1 | grep -v amdgpu.ids > $OUT/wave_sweep.txt
  WAVE_SWEEP=wave WAVE_PROF=1 python tools/wave_sweep.py 4 1024 2048 2>&1 | grep -v amdgpu.ids > $OUT/wave_phase_profile.txt
  for n in 1024 2048; do
    bash tools/profile_pmc.sh r06_$n nuts_wave tools/wave_run.py $n wave 500 4 > $OUT/wave_pmc_$n.log 2>&1
    grep -v "simple_timer\|amdgpu.ids" gpurun_out/pmc_r06_$n/summary.txt > $OUT/wave_sq_counters_$n.txt
  done
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06_wave -o run -- python3 tools/wave_run.py 2048 wave 1000 10 > $OUT/wave_trace.log 2>&1
  DB=$(find gpurun_out/prof_r06_wave -name '*.db' | head -1); [ -n "$DB" ] && python3 tools/rocpd_summary.py "$DB" > $OUT/wave_kernel_stats.txt
fi
if has soak; then
  python tools/soak_sampling.py 512 8 500 500 2>&1 | grep -v amdgpu.ids > $OUT/config4_end_to_end.txt
  python tools/soak_sampling.py 128 8 500 500 2>&1 | grep -v amdgpu.ids > $OUT/soak_128x8.txt
  python tools/soak_sampling.py 256 8 500 500 2>&1 | grep -v amdgpu.ids > $OUT/soak_256x8.txt
  python tools/soak_sampling.py 1536 8 200 200 2>&1 | grep -v amdgpu.ids > $OUT/oversubscribed.txt
  python tools/config3_run.py 2>&1 | grep -v amdgpu.ids > $OUT/config3.txt
fi
if has map; then
  python tools/map_timing.py > $OUT/map_timing.txt 2>&1
  python tools/map_suite_run.py > $OUT/map_suite.txt 2>&1
  python tools/map_suite_many.py 2>&1 | grep -v amdgpu.ids > $OUT/map_suite_many.txt
  python tools/lbfgs_pin.py 2>&1 | grep -v amdgpu.ids > $OUT/lbfgs_pin.txt
  python tools/lbfgs_cap_seeds.py 2>&1 | grep -v amdgpu.ids > $OUT/lbfgs_cap_seeds.txt
fi
if has config5; then
  python tools/bench_config5.py 4096 2>&1 | grep -v amdgpu.ids > $OUT/config5.txt                       # the rate (production kernel)
  python tools/bench_config5.py 4096 --series-outliers 2>&1 | grep -v amdgpu.ids >> $OUT/config5.txt
  echo '-- with the phase profile (profiling kernel, ~5 % slower):' >> $OUT/config5.txt
  python tools/bench_config5.py 4096 --phase-profile 2>&1 | grep -v amdgpu.ids >> $OUT/config5.txt
  python tools/config5_run.py 2>&1 | grep -v amdgpu.ids > $OUT/config5_run.txt
  # the same load with fewer workgroups on the chip (16 chains each): what the row passes cost without the other CUs' traffic
  for n in 2048 1024 256; do
    BDRT_CHAINS_PER_WG=16 BDRT_COMPACTION=0 BDRT_TAIL_MIGRATION=0 BDRT_WAVE=0 python tools/bench_config5.py $n --phase-profile 2>&1 | grep -v amdgpu.ids | head -6
  done > $OUT/config5_fewer_workgroups.txt
  # rocprofv3 records of nuts_kernel<27,4,0>: kernel trace, HBM counters (separate passes), SQ counters
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06_c5 -o run -- python3 tools/bench_config5.py 4096 > $OUT/config5_trace.log 2>&1
  DB=$(find gpurun_out/prof_r06_c5 -name '*.db' | head -1); [ -n "$DB" ] && python3 tools/rocpd_summary.py "$DB" > $OUT/config5_kernel_stats.txt
  : > $OUT/config5_pmc.txt
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $CTR -d gpurun_out/pmc_r06_c5_$CTR -o run -- python3 tools/bench_config5.py 4096 > $OUT/config5_pmc_$CTR.log 2>&1
    DB=$(find gpurun_out/pmc_r06_c5_$CTR -name '*.db' | head -1)
    [ -n "$DB" ] && python3 - "$DB" >> $OUT/config5_pmc.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for r in cur.execute("select counter_name, count(*), avg(value), max(value) from counters_collection where kernel_name like '%nuts_kernel%' group by counter_name"):
    print('%-12s dispatches %4d (50 rounds of 4096 chains each)  avg/dispatch %.6g  max %.6g   [raw counter units: KB on this image]' % r)
PY
    rm -f $OUT/config5_pmc_$CTR.log
  done
  bash tools/profile_pmc.sh r06_c5 nuts_kernel tools/bench_config5.py 4096 > $OUT/config5_sq.log 2>&1
  grep -v "simple_timer\|amdgpu.ids" gpurun_out/pmc_r06_c5/summary.txt > $OUT/config5_sq.txt
fi
if has waveom; then
  # the one-chain-per-wave kernel with the outlier models / several distributions against what those unit counts took before
  for fam in --series-outliers ""; do
    for n in 4 256 512 768 1024 1536; do
      python tools/bench_config5.py $n $fam 2>&1 | grep -v amdgpu.ids | head -1
      BDRT_WAVE=0 python tools/bench_config5.py $n $fam 2>&1 | grep -v amdgpu.ids | head -1 | sed "s/^/   BDRT_WAVE=0: /"
    done
  done > $OUT/wave_outliers_sweep.txt
  python tools/wave_outliers_study.py 100 2>&1 | grep -v amdgpu.ids > $OUT/wave_outliers_study.txt
fi
if has hmc; then
  python tools/hmc_suite_many.py 2>&1 | grep -v amdgpu.ids > $OUT/hmc_suite_many.txt
fi
if has fuzz; then
  python -m tests.fuzz_parity --first 4000 --count 300 > $OUT/fuzz_parity.txt 2>&1
  python -m tests.fuzz_inverter --first 4000 --count 100 > $OUT/fuzz_inverter.txt 2>&1
fi
# (what travels back is limited to 64 MiB: the rocprofv3 databases stay on the box, their summaries are in $OUT)
find gpurun_out -name '*.db' -delete 2>/dev/null
find gpurun_out -name '*.csv' -size +1M -delete 2>/dev/null
du -sh gpurun_out
ls -la $OUT
