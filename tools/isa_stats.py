#!/usr/bin/env python3
"""Per-function statistics of a hipcc -save-temps gfx950 .s file: instructions, v_mad_u64_u32, scratch / LDS / global accesses, calls, wait
states, registers and scratch frame.  Usage: python tools/isa_stats.py <file.s>"""
import re
import subprocess
import sys


def main(path):
    cur, stats, order = None, {}, []
    for l in open(path):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur = m.group(1)
            stats[cur] = dict(n=0, mad=0, addc=0, mov=0, st=0, ld=0, ds=0, gl=0, call=0, nop=0, vg=None, sc=None)
            order.append(cur)
            continue
        if cur is None:
            continue
        t = l.strip()
        m = re.match(r'; NumVgprs: (\d+)', t)
        if m:
            stats[cur]['vg'] = int(m.group(1))
        m = re.match(r'; ScratchSize: (\d+)', t)
        if m:
            stats[cur]['sc'] = int(m.group(1))
        if not t or t.startswith(('.', ';', '//')):
            continue
        d = stats[cur]
        d['n'] += 1
        for key, pre in (('mad', 'v_mad_u64_u32'), ('addc', ('v_addc', 'v_subb', 'v_add_co', 'v_sub_co')), ('mov', ('v_mov_b32', 'v_accvgpr')),
                         ('st', ('scratch_store', 'buffer_store')), ('ld', ('scratch_load', 'buffer_load')), ('ds', 'ds_'), ('gl', 'global_'),
                         ('call', 's_swappc'), ('nop', 's_nop')):
            if t.startswith(pre):
                d[key] += 1
    try:
        names = subprocess.run(['c++filt'], input="\n".join(order), capture_output=True, text=True).stdout.split("\n")
    except FileNotFoundError:
        names = order
    for k, nm in zip(order, names):
        print("%-100s %s" % (nm[:100], " ".join("%s=%s" % kv for kv in stats[k].items())))


if __name__ == "__main__":
    main(sys.argv[1])
