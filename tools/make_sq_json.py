"""SQ counters of the sampler kernel (tools/profile_sq.sh summary: counter, dispatches, average per dispatch) -> profiles/sq_counters.json:
what bench.py's `roofline.pipe_busy` is computed from.  pipe_busy = (SQ_VALU_MFMA_BUSY_CYCLES + 4 x (SQ_INSTS_VALU - SQ_INSTS_MFMA)) / SIMD cycles
of a launch: the fp64 pipe of a SIMD is occupied 4 cycles by a VALU instruction of a wave and for the counted cycles by an MFMA
(tools/ubench/f64_overlap.hip: the two do not overlap); SIMD cycles = 4 SIMDs x CUs x launch time x shader clock.
Usage: make_sq_json.py <summary.txt> <avg_launch_ms> <n_cu> <sclk_mhz> <source label>"""
import json
import sys

vals = {}
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) >= 5 and p[0].startswith('SQ_'):
        vals[p[0]] = float(p[-1])
ms, n_cu, mhz = float(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
simd_cycles = 4.0 * n_cu * ms * 1e-3 * mhz * 1e6
valu = vals['SQ_INSTS_VALU'] - vals.get('SQ_INSTS_MFMA', 0.0)
busy = vals['SQ_VALU_MFMA_BUSY_CYCLES'] + 4.0 * valu
print(json.dumps({'kernel': 'nuts_kernel', 'source': sys.argv[5], 'avg_launch_ms': ms, 'n_cu': n_cu, 'sclk_mhz': mhz, 'simd_cycles_per_launch': simd_cycles,
                  'counters_per_launch': vals, 'mfma_busy_frac': vals['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, 'valu_issue_frac': 4.0 * valu / simd_cycles,
                  'pipe_busy': busy / simd_cycles, 'wait_inst_any_over_wave_cycles': vals.get('SQ_WAIT_INST_ANY', 0.0) / max(vals.get('SQ_WAVE_CYCLES', 1.0), 1.0),
                  'note': 'rocprofv3 --pmc passes of `bench.py --gpus 1 --steps 20 --warmup 5` (tools/profile_sq.sh); averages over the 25 dispatches of the run'}, indent=1))
