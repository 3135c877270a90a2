#!/usr/bin/env python3
"""Static instruction mix of a kernel in the device ISA (hipcc -S --cuda-device-only output).

  python scripts/isa_stats.py /tmp/isa/cem.s 'cem_rollout_kernelILi1ELi1ELi0' [--blocks]

Prints, per basic block that holds MFMAs (or all of them with --blocks), the number of MFMA / VALU (by class) / LDS / VMEM /
SALU instructions.  Used to see what sits next to the fp32 MFMAs in the rollout kernels: every VALU instruction
there is paid in full.
"""
import collections
import re
import sys

TRANS = ('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32', 'v_rcp_iflag_f32')
QUARTER = ('v_mad_u64_u32', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_i64_i32', 'v_mul_hi_i32')


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith(TRANS):
        return 'valu_trans'
    if op.startswith(QUARTER):
        return 'valu_mul64'
    if op.startswith('v_pk_'):
        return 'valu_pk'
    if op.startswith(('v_permlane', 'v_readlane', 'v_readfirstlane', 'v_writelane')) or '_dpp' in op:
        return 'valu_xlane'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if not on:
            if ln.startswith('_Z') and name in ln.split(':')[0] and ln.rstrip().split(';')[0].strip().endswith(':'):
                on = True
            continue
        if ln.startswith('.Lfunc_end') or ln.lstrip().startswith('s_endpgm') and False:
            break
        out.append(ln.rstrip('\n'))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    all_blocks = '--blocks' in sys.argv
    lines = kernel_lines(path, name)
    blocks, cur, label = [], collections.Counter(), 'entry'
    ops = collections.Counter()
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith(';'):
            continue
        if re.match(r'^\.?[A-Za-z_][\w.$]*:', s):
            blocks.append((label, cur)); cur = collections.Counter(); label = s.split(':')[0]
            continue
        if s.startswith('.'):
            continue
        op = s.split()[0]
        c = classify(op)
        cur[c] += 1
        if c.startswith('valu'):
            ops[op.replace('_e32', '').replace('_e64', '')] += 1
    blocks.append((label, cur))
    tot = collections.Counter()
    for lb, c in blocks:
        tot.update(c)
        if all_blocks or c['mfma']:
            print('%-12s %s' % (lb, dict(sorted(c.items()))))
    print('TOTAL', dict(sorted(tot.items())))
    print('VALU ops:', ', '.join('%s %d' % kv for kv in ops.most_common(40)))


if __name__ == '__main__':
    main()
