#!/usr/bin/env python3
"""Instruction statistics of one kernel in a hipcc -S listing.

    python tools/isa_stats.py file.s <mangled-name-substring> [--loop]

Prints VGPR / SGPR / scratch / LDS of the kernel and the number of VALU, SALU, LDS (ds_), VMEM, MFMA,
waitcnt, barrier and branch instructions, for the whole kernel and (with --loop) per basic block, so that
the body of the candidate loop can be compared between builds without a GPU.
"""
import re
import sys


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return 'mfma'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, pat = sys.argv[1], sys.argv[2]
    per_block = '--loop' in sys.argv
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_Z') and pat in l and l.rstrip().split(':')[0].endswith(pat.split()[-1]) or (l.startswith('_Z') and pat in l and ':' in l):
            start = i
            break
    if start is None:
        sys.exit('kernel not found')
    name = lines[start].split(':')[0]
    tot, blocks, cur = {}, [], None
    meta = {}
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.end_amdhsa_kernel') or s.startswith('.Lfunc_end'):
            if s.startswith('.Lfunc_end'):
                pass
        m = re.match(r'\.(?:amdhsa_next_free_vgpr|amdhsa_next_free_sgpr|amdhsa_group_segment_fixed_size|amdhsa_private_segment_fixed_size|amdhsa_accum_offset)\s+(\S+)', s)
        if m:
            meta[s.split()[0]] = m.group(1)
        if s.startswith('.end_amdhsa_kernel'):
            break
        if s.startswith('; ') and ('NumVgprs' in s or 'ScratchSize' in s or 'Occupancy' in s or 'NumAgprs' in s):
            meta[s[2:].split(':')[0]] = s.split(':')[1].strip()
        if re.match(r'^\.LBB\d+_\d+:', s):
            cur = [s.split(':')[0], {}]
            blocks.append(cur)
            continue
        if not s or s.startswith((';', '.', '//')):
            continue
        op = s.split()[0]
        c = classify(op)
        tot[c] = tot.get(c, 0) + 1
        if cur is not None:
            cur[1][c] = cur[1].get(c, 0) + 1
    print(name[:100])
    print('  meta:', meta)
    print('  total:', dict(sorted(tot.items())))
    if per_block:
        for nm, d in blocks:
            n = sum(d.values())
            if n >= 40:
                print('  %-12s %5d  %s' % (nm, n, dict(sorted(d.items()))))


if __name__ == '__main__':
    main()
