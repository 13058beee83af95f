"""Basic-block instruction mix of one kernel in /tmp/kernels.s (largest blocks first)."""
import re, sys, collections
pat = sys.argv[1]
minsz = int(sys.argv[2]) if len(sys.argv) > 2 else 60
txt = open('/tmp/kernels.s').read().split('\n')
start = next(i for i, l in enumerate(txt) if l.startswith(pat) and l.rstrip().endswith(': ; @' + pat.rstrip(':')) or l.startswith(pat + ':'))
end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in txt[start + 1:end]:
    m = re.match(r'^(\.LBB[0-9_]+):', l)
    if m:
        blocks.append((name, cur)); cur = []; name = m.group(1); continue
    m = re.match(r'^\s+([a-z_0-9]+)\s', l)
    if m: cur.append(m.group(1))
blocks.append((name, cur))
def cls(op):
    if op.endswith('_f64') or '_f64_' in op: return 'f64'
    if op.startswith('v_') and 'dpp' in op: return 'dpp'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_load'): return 'gload'
    if op.startswith('global_store'): return 'gstore'
    if op.startswith('scratch'): return 'scratch'
    if 'accvgpr' in op: return 'acc'
    if op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane'): return 'lane'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_'): return 'salu'
    return 'other'
tot = collections.Counter()
for name, ops in blocks:
    c = collections.Counter(cls(o) for o in ops)
    tot.update(c)
    if len(ops) >= minsz:
        print(f"{name:14s} n={len(ops):5d} ", ' '.join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
print('TOTAL', sum(tot.values()), dict(tot))
