"""Static per-function statistics of the decision kernel's gfx950 assembly (tools/cc_rd.sh leaves it in /tmp): instruction count, scratch
loads / stores (register saves and spills), waits, cross-lane reads, LDS / global operations, calls."""
import re, collections, subprocess, sys
S = sys.argv[1] if len(sys.argv) > 1 else '/tmp/rd_kernel-hip-amdgcn-amd-amdhsa-gfx950.s'
cur = None; stats = collections.OrderedDict()
for l in open(S):
    m = re.match(r'^(_Z\w+|hevcdl\w+):', l)
    if m: cur = m.group(1); stats[cur] = collections.Counter(); continue
    if cur is None: continue
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')):
        if t.startswith('.Lfunc_end'): cur = None
        continue
    op = t.split()[0]; c = stats[cur]
    c['n'] += 1
    if op.startswith('scratch_load'): c['sl'] += 1
    if op.startswith('scratch_store'): c['ss'] += 1
    if op.startswith('s_waitcnt'): c['w'] += 1
    if op.startswith('v_readlane') or op.startswith('v_readfirstlane'): c['rl'] += 1
    if op.startswith('ds_'): c['ds'] += 1
    if op.startswith('global_'): c['gl'] += 1
    if op.startswith('s_swappc'): c['call'] += 1
names = subprocess.run(['c++filt'], input='\n'.join(stats), capture_output=True, text=True).stdout.splitlines()
tot = collections.Counter()
for (k, v), nm in sorted(zip(stats.items(), names), key=lambda kv: -kv[0][1]['n']):
    tot.update(v)
    print("%6d ins  sl %4d ss %4d wait %4d rl %4d ds %4d gl %4d call %3d  %s" % (v['n'], v['sl'], v['ss'], v['w'], v['rl'], v['ds'], v['gl'], v['call'], re.sub(r'\(anonymous namespace\)::', '', nm)[:100]))
print("TOTAL %d ins, scratch loads %d stores %d" % (tot['n'], tot['sl'], tot['ss']))
