import re, sys
lines = open(sys.argv[1]).read().split('\n')
start, end = int(sys.argv[2]), int(sys.argv[3])
blocks = []  # (label, first_line, insts)
cur = ('entry', start, [])
for n in range(start, end):
    l = lines[n]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = (m.group(1), n + 1, [])
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur[2].append(t)
blocks.append(cur)
idx = {b[0]: i for i, b in enumerate(blocks)}
# back edges
loops = []
for i, b in enumerate(blocks):
    for ins in b[2]:
        m = re.match(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', ins)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in idx and idx[tgt] <= i: loops.append((idx[tgt], i))
depth = [0] * len(blocks)
for a, b in loops:
    for i in range(a, b + 1): depth[i] += 1
tot = {}
for i, b in enumerate(blocks):
    v = sum(1 for x in b[2] if x.startswith('v_') and not x.startswith('v_readlane') and not x.startswith('v_writelane') and not x.startswith('v_readfirstlane'))
    rl = sum(1 for x in b[2] if x.startswith('v_readlane')); wl = sum(1 for x in b[2] if x.startswith('v_writelane')); rf = sum(1 for x in b[2] if x.startswith('v_readfirstlane'))
    s = sum(1 for x in b[2] if x.startswith('s_') and not x.startswith('s_waitcnt') and not x.startswith('s_nop'))
    mem = sum(1 for x in b[2] if x.startswith('global_') or x.startswith('buffer_') or x.startswith('scratch_') or x.startswith('flat_'))
    sc = sum(1 for x in b[2] if x.startswith('scratch_'))
    lds = sum(1 for x in b[2] if x.startswith('ds_'))
    print(f"{b[0]:12s} line {b[1]:5d} depth {depth[i]} valu {v:3d} readlane {rl:2d} writelane {wl:2d} rfl {rf:2d} salu {s:3d} mem {mem:2d} scratch {sc:2d} lds {lds:2d}")
    d = depth[i]
    t = tot.setdefault(d, [0,0,0,0,0]); t[0]+=v; t[1]+=rl; t[2]+=wl; t[3]+=s; t[4]+=sc
print(tot)
