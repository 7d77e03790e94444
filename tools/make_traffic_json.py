"""HBM traffic of the sampler kernel from the two rocprofv3 PMC passes of tools/profile_bench.sh (FETCH_SIZE and WRITE_SIZE in
separate runs, as the MI355X guide prescribes), per launch and per leapfrog round.  Units: the counters are in KB; on gfx950
FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, so `corrected` doubles it (our state-vector loads are
8 B/lane and the true value lies between raw and corrected)."""
import glob, json, os, sqlite3, sys

out = sys.argv[1]
vals = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    db = glob.glob(os.path.join(out, 'pmc_' + c, '**', '*.db'), recursive=True)
    if not db:
        continue
    cur = sqlite3.connect(db[0]).cursor()
    rows = cur.execute("select value from counters_collection where kernel_name like '%nuts_kernel%' and counter_name = ? "
                       "order by start", (c,)).fetchall()
    allv = [r[0] for r in rows]
    v = allv[-20:]                                  # the 20 timed launches (the first 5 are warm-up: chains still in their
    vals[c] = sum(v) / max(len(v), 1)               #  step-size search touch fewer rows, which is why the 25-launch average
    vals[c + '_all'] = sum(allv) / max(len(allv), 1)   # printed in pmc_<counter>.txt differs by a percent or two)
    vals[c + '_n'] = len(allv)
line = json.load(open(os.path.join(out, 'bench_line.json')))
rounds = line['config']['rounds_per_launch']
raw = (vals.get('FETCH_SIZE', 0) + vals.get('WRITE_SIZE', 0)) * 1024
cor = (2 * vals.get('FETCH_SIZE', 0) + vals.get('WRITE_SIZE', 0)) * 1024
print(json.dumps({'kernel': 'nuts_kernel', 'rounds_per_launch': rounds, 'units': line['config']['units_per_gpu'],
                  'FETCH_SIZE_KB_per_launch': vals.get('FETCH_SIZE'), 'WRITE_SIZE_KB_per_launch': vals.get('WRITE_SIZE'),
                  'FETCH_SIZE_KB_per_launch_all_dispatches': vals.get('FETCH_SIZE_all'),
                  'WRITE_SIZE_KB_per_launch_all_dispatches': vals.get('WRITE_SIZE_all'),
                  'dispatches': vals.get('FETCH_SIZE_n'),
                  'source': 'profiles/%s: pmc_FETCH_SIZE.txt / pmc_WRITE_SIZE.txt (average of the last 20 of the dispatches listed there = the timed launches)' % os.path.basename(out.rstrip('/')).replace('prof_', ''),
                  'hbm_bytes_per_launch_raw': raw, 'hbm_bytes_per_launch_corrected': cor,
                  'hbm_bytes_per_round_corrected': cor / rounds, 'hbm_bytes_per_round_raw': raw / rounds,
                  'algorithmic_bytes_per_round': 830760 + 7888 * line['config']['units_per_gpu'],
                  'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024'},
                 indent=1))
