"""Instruction census of a kernel's hot loop from `hipcc --cuda-device-only -S` output.
usage: loop_census.py file.s <substring of the kernel symbol> [<substring that marks the hot loop, default global_load_dwordx4>]
The hot loop = the innermost labelled loop (".LBBn_m: ... s_cbranch* .LBBn_m" back edge) whose body holds the most marker lines."""
import re, sys
path, sym = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else 'global_load_dwordx4'
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sym in l and ':' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def nmark(a, b): return sum(marker in x for x in body[a:b + 1])
best = max(loops, key=lambda ab: (nmark(*ab), -(ab[1] - ab[0])))
a, b = best
ins = [x.strip() for x in body[a:b + 1] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
def cnt(pred): return sum(1 for x in ins if pred(x))
print("kernel %s: hot loop lines %d..%d, %d instructions" % (sym, a, b, len(ins)))
print("  vector (v_*) %d, of which v_readlane %d, v_writelane %d" % (cnt(lambda x: x.startswith('v_')), cnt(lambda x: x.startswith('v_readlane')), cnt(lambda x: x.startswith('v_writelane'))))
print("  scalar (s_*) %d, of which s_load %d, s_waitcnt %d, branches %d" % (cnt(lambda x: x.startswith('s_')), cnt(lambda x: x.startswith('s_load')), cnt(lambda x: x.startswith('s_waitcnt')), cnt(lambda x: x.startswith(('s_cbranch', 's_branch')))))
print("  global loads %d, stores %d, LDS (ds_*) %d" % (cnt(lambda x: x.startswith('global_load')), cnt(lambda x: x.startswith('global_store')), cnt(lambda x: x.startswith('ds_'))))
