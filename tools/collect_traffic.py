#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE -- separate runs, as MI355X_MICROARCH.md prescribes)
into profiles/r2_traffic.json: HBM-side bytes per launch of the dominant kernel.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 'conv_gemm_ring_kernel<unsigned short, 256, 256' 'conv_gemm_ring_kernel<bf16, 256, 256>'

Units/corrections: FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
(16 B/lane) coalesced streams -- which is what global_load_lds issues -- at 64 B, so the read side is doubled.  WRITE_SIZE was
checked against a known byte count here (PPM conv: 302 MiB expected, 295424 KiB reported): no correction."""
import glob, hashlib, json, os, re, sqlite3, sys
from collections import defaultdict

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'segland_amd', 'csrc', 'libsegland_hip.so')


def per_launch(path, counter, needle):
    db = sqlite3.connect(glob.glob(path + '/**/*.db', recursive=True)[0])
    tot, disp = 0.0, set()
    for name, did, cn, cv in db.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
        if cn == counter and needle in name:
            tot += cv; disp.add(did)
    return tot / max(len(disp), 1), len(disp)


fetch_dir, write_dir, needle, label = sys.argv[1:5]
f_kib, nf = per_launch(fetch_dir, 'FETCH_SIZE', needle)
w_kib, nw = per_launch(write_dir, 'WRITE_SIZE', needle)
out = {'kernel': label, 'launches_profiled': [nf, nw], 'fetch_kib_per_launch_raw': f_kib, 'write_kib_per_launch_raw': w_kib,
       'fetch_correction': 2.0, 'hbm_bytes_per_launch': int(f_kib * 1024 * 2.0 + w_kib * 1024),
       'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 3 --warmup 1',
       'lib_sha256_16': hashlib.sha256(open(LIB, 'rb').read()).hexdigest()[:16]}      # bench.py quotes these bytes only next to the library they were measured with
json.dump(out, open(sys.argv[5] if len(sys.argv) > 5 else 'profiles/r2_traffic.json', 'w'), indent=1)
print(json.dumps(out))
