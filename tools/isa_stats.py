"""Static instruction mix of one kernel in the gfx950 assembly `make -C csrc asm` writes. usage: isa_stats.py <file.s> <substring of the mangled name> [-v]"""
import re, sys
from collections import Counter
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    if re.match(r'^_Z\S*:', l) and pat in l.split(':')[0]:
        start = i
        break
assert start is not None, "kernel not found"
ins = []
for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end'):
        break
    t = l.strip()
    if not l.startswith('\t') or not t or t[0] in '.;':
        continue
    ins.append(t)
c = Counter(t.split()[0] for t in ins)
def grp(pred):
    return sum(v for k, v in c.items() if pred(k))
print(lines[start].split(':')[0][:80])
print(" total", len(ins), " valu", grp(lambda k: k.startswith('v_')), " salu", grp(lambda k: k.startswith('s_') and not k.startswith('s_load') and not k.startswith('s_waitcnt')),
      " smem", grp(lambda k: k.startswith('s_load')), " vmem", grp(lambda k: k.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))), " lds", grp(lambda k: k.startswith('ds_')),
      " waitcnt", c.get('s_waitcnt', 0), " barrier", c.get('s_barrier', 0))
print(" f64", grp(lambda k: k.endswith('_f64') or '_f64_' in k), " f32", grp(lambda k: k.endswith('_f32') or '_f32_' in k), " readlane/writelane", c.get('v_readlane_b32', 0) + c.get('v_writelane_b32', 0),
      " scratch", grp(lambda k: k.startswith('scratch_')), " mov", c.get('v_mov_b32_e32', 0) + c.get('v_mov_b32_dpp', 0), " cndmask", grp(lambda k: k.startswith('v_cndmask')),
      " cmp", grp(lambda k: k.startswith('v_cmp')), " branch", grp(lambda k: k.startswith('s_cbranch') or k == 's_branch'), " mfma", grp(lambda k: 'mfma' in k))
if '-v' in sys.argv:
    for k, v in c.most_common(40):
        print("   %-28s %d" % (k, v))
