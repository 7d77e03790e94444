#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06p
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r06p/p -o run -- python3 tools/dbg/map_once.py > gpurun_out/r06p/log.txt 2>&1
DB=$(find gpurun_out/r06p/p -name '*.db' | head -1)
python3 - "$DB" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
print([r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")][:60])
PY
python3 tools/dbg/rocpd_timeline.py "$DB" 2400 40 > gpurun_out/r06p/timeline.txt 2>&1
cat gpurun_out/r06p/timeline.txt
find gpurun_out/r06p -name '*.db' -delete
