"""Developer tool: instruction mix per basic block of one kernel in a gfx950 .s file.
python tools/isa_mix.py file.s <mangled-name substring>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*%s\S*):[^\n]*\n(.*?)\n\s*s_endpgm' % re.escape(key), s, re.S | re.M)
print(m.group(1))
lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith(';')]
blk = 'entry'
blocks = collections.OrderedDict({blk: []})
for l in lines:
    if l.startswith('.LBB'):
        blk = l.split(':')[0]
        blocks[blk] = []
        continue
    blocks[blk].append(l)
for b, ls in blocks.items():
    c = collections.Counter()
    for l in ls:
        op = l.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_accvgpr'): c['acc'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('s_waitcnt'): c['wait'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_nop'): c['nop'] += 1
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): c['br'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        else: c[op] += 1
    print(b, len(ls), dict(c))
i = s.find('.name:', s.find('amdhsa.kernels'))
for blk in s[s.find('amdhsa.kernels'):].split('  - .agpr_count')[1:]:
    if key in blk:
        for k in ['.agpr_count', '.vgpr_count', '.sgpr_count', '.vgpr_spill_count', '.group_segment_fixed_size']:
            mm = re.search(r'%s:\s*(\d+)' % re.escape(k), '.agpr_count' + blk)
            print(k, mm.group(1) if mm else None)
