#!/usr/bin/env python3
"""Per-kernel-family counter evidence from rocprofv3 --pmc passes over bench.py (kernel by kernel, --no-step-graph): for every kernel family above `min_ms` per step

  pass A  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE   -> MFMA-busy share of the SIMDs' busy time (MFMA_BUSY / (32 x SQ_BUSY_CYCLES), the convention of
                                                                        profiles/r2_pmc_conv.txt), of all SIMD time (MFMA_BUSY x 8 / (1024 x GRBM_GUI_ACTIVE)), and the sustained
                                                                        shader clock GRBM_GUI_ACTIVE / (8 x kernel duration).  rocprofv3 reports GRBM_GUI_ACTIVE summed over the
                                                                        8 XCDs (16-18 "GHz" undivided); it also counts active cycles around the kernel proper, so the clock is
                                                                        given only for kernels of >= 100 us (5-10 % high at 100-200 us)
  pass B  SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pass C / D  FETCH_SIZE / WRITE_SIZE (separate passes; FETCH_SIZE doubled: MI355X_MICROARCH.md)   -> HBM-side bytes per launch and TB/s

usage: pmc_families.py [--min-ms 0.25] [--cmd '<profiled command>'] <out.txt> <out.json> <steps> <dirA> [<dirB> [<dirC> <dirD>]]"""
import glob, hashlib, json, os, re, sqlite3, sys
from collections import defaultdict

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'segland_amd', 'csrc', 'libsegland_hip.so')


def load(path):
    db = sqlite3.connect(glob.glob(path + '/**/*.db', recursive=True)[0])
    per = defaultdict(lambda: defaultdict(float)); dur = defaultdict(dict)
    for name, did, d, cn, cv in db.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events"):
        n = re.sub(r'\(anonymous namespace\)::', '', name); n = re.sub(r'^void ', '', n); n = re.sub(r'\(.*', '', n)
        per[n][cn] += cv; dur[n][did] = d
    return per, dur


min_ms, cmd = 0.25, 'python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-step-graph --no-other-configs'
args = sys.argv[1:]
while args and args[0].startswith('--'):              # --min-ms X: families below X ms per step are left out;  --cmd '...': the profiled command, for the header
    if args[0] == '--min-ms': min_ms = float(args[1])
    elif args[0] == '--cmd': cmd = args[1]
    args = args[2:]
out_txt, out_json, steps = args[0], args[1], int(args[2])
dirs = args[3:]
A, durA = load(dirs[0])
B, durB = load(dirs[1]) if len(dirs) > 1 else ({}, {})
C, durC = load(dirs[2]) if len(dirs) > 3 else ({}, {})
D, durD = load(dirs[3]) if len(dirs) > 3 else ({}, {})
lines = ['# rocprofv3 --kernel-trace --pmc <counters> -- %s (one pass per counter set);' % cmd,
         '# per kernel name over all its launches in the run; ms/step from the counter pass itself (PMC passes serialise kernels: durations are close to, not equal to, the un-profiled ones)',
         '%-62s %6s %8s %9s %9s %8s %9s %9s %8s %8s %9s' % ('kernel', 'calls', 'ms/step', 'mfma/busy', 'mfma/all', 'clk GHz', 'lds confl', 'wait_inst', 'active', 'MB/launch', 'TB/s')]
js = {}
for n in sorted(A, key=lambda k: -sum(durA[k].values())):
    calls = len(durA[n]); ms_step = sum(durA[n].values()) / 1e6 / steps
    if ms_step < min_ms:
        continue
    a = A[n]
    busy, mf, gui = a.get('SQ_BUSY_CYCLES', 0), a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), a.get('GRBM_GUI_ACTIVE', 0)
    t_ns = sum(durA[n].values())
    r = {'calls': calls, 'ms_per_step': round(ms_step, 3),
         'mfma_busy_over_sq_busy': round(mf / (32 * busy), 4) if busy else None,
         'mfma_busy_over_all_simd_cycles': round(mf * 8 / (1024 * gui), 4) if gui else None,
         'clock_ghz': round(gui / 8 / t_ns, 3) if gui and t_ns and t_ns / calls >= 1e5 else None,
         'avg_us': round(t_ns / calls / 1e3, 1)}
    b = B.get(n, {})
    if b.get('SQ_LDS_IDX_ACTIVE'):
        r['lds_conflict_share'] = round(b.get('SQ_LDS_BANK_CONFLICT', 0) / b['SQ_LDS_IDX_ACTIVE'], 3)
    if b.get('SQ_WAVE_CYCLES'):
        r['wait_inst_share'] = round(b.get('SQ_WAIT_INST_ANY', 0) / b['SQ_WAVE_CYCLES'], 3)
        r['active_inst_share'] = round(b.get('SQ_ACTIVE_INST_ANY', 0) / b['SQ_WAVE_CYCLES'], 3)
    if n in C and n in D and durC.get(n) and durD.get(n):
        fb = C[n].get('FETCH_SIZE', 0) * 1024 * 2.0 / len(durC[n]); wb = D[n].get('WRITE_SIZE', 0) * 1024 / len(durD[n])
        r['hbm_mb_per_launch'] = round((fb + wb) / 1e6, 1)
        r['hbm_tb_per_s'] = round((fb + wb) / (sum(durC[n].values()) / len(durC[n])) / 1e3, 2)       # bytes / ns = GB/s -> /1e3 TB/s
    js[n] = r
    f = lambda k, fmt: (fmt % r[k]) if r.get(k) is not None else '-'
    lines.append('%-62s %6d %8.3f %9s %9s %8s %9s %9s %8s %9s %9s' % (n[:62], calls, ms_step, f('mfma_busy_over_sq_busy', '%.3f'), f('mfma_busy_over_all_simd_cycles', '%.3f'), f('clock_ghz', '%.2f'),
                 f('lds_conflict_share', '%.3f'), f('wait_inst_share', '%.3f'), f('active_inst_share', '%.3f'), f('hbm_mb_per_launch', '%.1f'), f('hbm_tb_per_s', '%.2f')))
open(out_txt, 'w').write('\n'.join(lines) + '\n')
json.dump({'lib_sha256_16': hashlib.sha256(open(LIB, 'rb').read()).hexdigest()[:16], 'command': cmd, 'kernels': js}, open(out_json, 'w'), indent=1)      # bench.py reads `kernels` only when the sha matches the loaded library
print('\n'.join(lines[:40]))
