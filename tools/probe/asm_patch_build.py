"""Build ONE .hip source with a patch applied to its gfx950 ASSEMBLY (r06: bisecting the SLP / packed-f32 miscount of the
projection epilogue of lp_hi_stream.hip at instruction level).

    python tools/probe/asm_patch_build.py SRC.hip OUT.o PATCH [extra hipcc flags ...]

Replays hipcc's own sub-commands (`hipcc -###  -save-temps`): device compile to .s, THEN `PATCH` (a name from PATCHES below)
rewrites the .s, then assembler, lld, offload bundler and the host compile that embeds the bundle.  The object is linked into
a copy of libkge_hip.so by the caller (tools/probe/slp_bisect.sh).
"""
import os
import re
import shlex
import subprocess
import sys
import tempfile

HIPCC = '/opt/rocm/bin/hipcc'
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DKGE_BUILD_NO_SLP=1"]


def _pk_fma_sgpr_to_scalar(lines):
    """v_pk_fma_f32 vD, vA, s[N:N+1], vC op_sel_hi:[1,0,1]  ->  two v_fma_f32 with the SGPR's low dword (same values)."""
    out, n = [], 0
    pat = re.compile(r'^\s*v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], s\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel_hi:\[1,0,1\]\s*$')
    for l in lines:
        m = pat.match(l)
        if m:
            d0, d1, a0, a1, s0, _, c0, c1 = map(int, m.groups())
            out.append('\tv_fma_f32 v%d, v%d, s%d, v%d' % (d0, a0, s0, c0))
            out.append('\tv_fma_f32 v%d, v%d, s%d, v%d' % (d1, a1, s0, c1))
            n += 1
        else:
            out.append(l)
    return out, n


def _pk_fma_opsel_to_scalar(lines):
    """v_pk_fma_f32 vD, vA, v[B:B+1], v[B:B+1] op_sel:[0,1,0] op_sel_hi:[1,1,0]  (fma(x, z, p), (p, z) in ONE pair) -> two v_fma_f32."""
    out, n = [], 0
    pat = re.compile(r'^\s*v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,0\] op_sel_hi:\[1,1,0\]\s*$')
    for l in lines:
        m = pat.match(l)
        if m:
            d0, d1, a0, a1, b0, b1, c0, c1 = map(int, m.groups())
            assert (b0, b1) == (c0, c1)
            # the destination pair may overlap a source pair: go through the order that reads before it writes
            assert d0 not in (a1, b0, b1), l
            out.append('\tv_fma_f32 v%d, v%d, v%d, v%d' % (d0, a0, b1, b0))
            out.append('\tv_fma_f32 v%d, v%d, v%d, v%d' % (d1, a1, b1, b0))
            n += 1
        else:
            out.append(l)
    return out, n


def _pk_mul_to_scalar(lines):
    out, n = [], 0
    pat = re.compile(r'^\s*v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]\s*$')
    for l in lines:
        m = pat.match(l)
        if m:
            d0, d1, a0, a1, b0, b1 = map(int, m.groups())
            assert d0 not in (a1, b1), l
            out.append('\tv_mul_f32_e32 v%d, v%d, v%d' % (d0, a0, b0))
            out.append('\tv_mul_f32_e32 v%d, v%d, v%d' % (d1, a1, b1))
            n += 1
        else:
            out.append(l)
    return out, n


def _nop_before_pk(lines):
    """two s_nop 15 in front of every packed f32 instruction (a missing wait state would be covered)"""
    out, n = [], 0
    for l in lines:
        if re.match(r'^\s*v_pk_(fma|mul|add)_f32 ', l):
            out += ['\ts_nop 15', '\ts_nop 15']
            n += 1
        out.append(l)
    return out, n


def _nop_after_pk(lines):
    out, n = [], 0
    for l in lines:
        out.append(l)
        if re.match(r'^\s*v_pk_(fma|mul|add)_f32 ', l):
            out += ['\ts_nop 15']
            n += 1
    return out, n


_OPSEL = re.compile(r'^\s*v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,0\] op_sel_hi:\[1,1,0\]\s*$')


def _bump_vgprs(lines, n=248):
    """the patches below use v[240:247] as scratch: every kernel descriptor of the file gets room for them"""
    out = []
    for l in lines:
        m = re.match(r'^(\s*\.amdhsa_next_free_vgpr )(\d+)\s*$', l)
        if m and int(m.group(2)) < n:
            l = m.group(1) + str(n)
        m = re.match(r'^(\s*\.vgpr_count:\s*)(\d+)\s*$', l)
        if m and int(m.group(2)) < n:
            l = m.group(1) + str(n)
        out.append(l)
    return out


def _opsel_to_plain_pairs(lines):
    """the op_sel form -> the PLAIN form on freshly materialised {z, z} / {p, p} pairs (what hipcc itself emits further down)"""
    out, n = [], 0
    for l in lines:
        m = _OPSEL.match(l)
        if m:
            d0, d1, a0, a1, b0, b1, _, _ = map(int, m.groups())
            out += ['\tv_mov_b32_e32 v240, v%d' % b1, '\tv_mov_b32_e32 v241, v%d' % b1, '\tv_mov_b32_e32 v242, v%d' % b0,
                    '\tv_mov_b32_e32 v243, v%d' % b0,
                    '\tv_pk_fma_f32 v[%d:%d], v[%d:%d], v[240:241], v[242:243]' % (d0, d1, a0, a1)]
            n += 1
        else:
            out.append(l)
    return _bump_vgprs(out), n


def _opsel_two_copies(lines):
    """the op_sel form kept, but src1 and src2 read two COPIES of the (p, z) pair instead of the same registers"""
    out, n = [], 0
    for l in lines:
        m = _OPSEL.match(l)
        if m:
            d0, d1, a0, a1, b0, b1, _, _ = map(int, m.groups())
            out += ['\tv_mov_b32_e32 v240, v%d' % b0, '\tv_mov_b32_e32 v241, v%d' % b1, '\tv_mov_b32_e32 v242, v%d' % b0,
                    '\tv_mov_b32_e32 v243, v%d' % b1,
                    '\tv_pk_fma_f32 v[%d:%d], v[%d:%d], v[240:241], v[242:243] op_sel:[0,1,0] op_sel_hi:[1,1,0]' % (d0, d1, a0, a1)]
            n += 1
        else:
            out.append(l)
    return _bump_vgprs(out), n


def _opsel_src0_copy(lines):
    """the op_sel form kept on the same (p, z) pair, but src0 (the just-loaded X values) read from a copy"""
    out, n = [], 0
    for l in lines:
        m = _OPSEL.match(l)
        if m:
            d0, d1, a0, a1, b0, b1, _, _ = map(int, m.groups())
            out += ['\tv_mov_b32_e32 v240, v%d' % a0, '\tv_mov_b32_e32 v241, v%d' % a1,
                    '\tv_pk_fma_f32 v[%d:%d], v[240:241], v[%d:%d], v[%d:%d] op_sel:[0,1,0] op_sel_hi:[1,1,0]' % (d0, d1, b0, b1, b0, b1)]
            n += 1
        else:
            out.append(l)
    return _bump_vgprs(out), n


def _opsel_hybrid(kind):
    """the op_sel form kept; kind 'fix_hi' / 'fix_lo': ONE lane of its result recomputed by a scalar v_fma_f32 afterwards;
    'src1_only': only src1 (z) selected by op_sel, src2 a plain {p, p} pair; 'src2_only': src1 a plain {z, z} pair, only src2 (p)
    selected by op_sel_hi"""
    def run(lines):
        out, n = [], 0
        for l in lines:
            m = _OPSEL.match(l)
            if not m:
                out.append(l)
                continue
            d0, d1, a0, a1, b0, b1, _, _ = map(int, m.groups())
            n += 1
            if kind == 'fix_hi':
                assert d1 not in (a1, b0, b1) and d0 not in (a1, b0, b1)
                out += [l, '\tv_fma_f32 v%d, v%d, v%d, v%d' % (d1, a1, b1, b0)]
            elif kind == 'fix_lo':
                assert d0 not in (a0, b0, b1) and d1 not in (a0, b0, b1)
                out += [l, '\tv_fma_f32 v%d, v%d, v%d, v%d' % (d0, a0, b1, b0)]
            elif kind == 'src1_only':
                out += ['\tv_mov_b32_e32 v242, v%d' % b0, '\tv_mov_b32_e32 v243, v%d' % b0,
                        '\tv_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[242:243] op_sel:[0,1,0] op_sel_hi:[1,1,1]' % (d0, d1, a0, a1, b0, b1)]
            elif kind == 'src2_only':
                out += ['\tv_mov_b32_e32 v240, v%d' % b1, '\tv_mov_b32_e32 v241, v%d' % b1,
                        '\tv_pk_fma_f32 v[%d:%d], v[%d:%d], v[240:241], v[%d:%d] op_sel_hi:[1,1,0]' % (d0, d1, a0, a1, b0, b1)]
        return _bump_vgprs(out), n
    return run


def _opsel_nop_only(lines):
    """the op_sel form untouched, eight s_nop 15 in front of each (1000+ cycles: any fixed-latency hazard would be covered)"""
    out, n = [], 0
    for l in lines:
        if _OPSEL.match(l):
            out += ['\ts_nop 15'] * 8
            n += 1
        out.append(l)
    return out, n


def _chain(*fs):
    def run(lines):
        tot = 0
        for f in fs:
            lines, n = f(lines)
            tot += n
        return lines, tot
    return run


PATCHES = {
    'none': lambda ls: (ls, 0),
    'sgpr_fma_scalar': _pk_fma_sgpr_to_scalar,          # only the fmaf(corr, -2^23, acc) with the SGPR-pair operand
    'opsel_fma_scalar': _pk_fma_opsel_to_scalar,        # only the fmaf(x, z, p) whose (p, z) sit in one register pair
    'mul_scalar': _pk_mul_to_scalar,                    # only corr = x * (...)
    'all_scalar': _chain(_pk_fma_sgpr_to_scalar, _pk_fma_opsel_to_scalar, _pk_mul_to_scalar),
    'nop_before_pk': _nop_before_pk,
    'nop_after_pk': _nop_after_pk,
    'opsel_plain_pairs': _opsel_to_plain_pairs,
    'opsel_two_copies': _opsel_two_copies,
    'opsel_src0_copy': _opsel_src0_copy,
    'opsel_nop8': _opsel_nop_only,
    'opsel_fix_hi': _opsel_hybrid('fix_hi'),
    'opsel_fix_lo': _opsel_hybrid('fix_lo'),
    'opsel_src1_only': _opsel_hybrid('src1_only'),
    'opsel_src2_only': _opsel_hybrid('src2_only'),
}


def main():
    src, out, patch = sys.argv[1], os.path.abspath(sys.argv[2]), sys.argv[3]
    extra = sys.argv[4:]
    work = tempfile.mkdtemp(prefix='asm_patch_')
    cmd = [HIPCC] + FLAGS + extra + ['-c', os.path.abspath(src), '-o', out, '-save-temps', '-###']
    txt = subprocess.run(cmd, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    steps = [l.strip() for l in txt.split('\n') if l.startswith(' "')]
    assert steps, txt
    patched = None
    for st in steps:
        argv = shlex.split(st)
        subprocess.run(argv, cwd=work, check=True)
        o = argv[argv.index('-o') + 1] if '-o' in argv else ''
        if o.endswith('-gfx950.s') and patched is None:
            p = os.path.join(work, o)
            lines = open(p).read().split('\n')
            lines, n = PATCHES[patch](lines)
            open(p, 'w').write('\n'.join(lines))
            patched = n
    print('%s: patch %s rewrote %d instructions -> %s' % (os.path.basename(src), patch, patched, out))


if __name__ == '__main__':
    main()
