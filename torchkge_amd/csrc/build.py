# -*- coding: utf-8 -*-
"""Build libkge_hip.so (gfx950) in-tree with hipcc.

    python -m torchkge_amd.csrc.build          # or __graft_entry__.build()

hipcc cross-compiles for gfx950 without a GPU.  The shared object lands next to
the sources (torchkge_amd/csrc/libkge_hip.so): git-ignored, but it travels to
the GPU box with the gpurun snapshot.  -ffp-contract=off is part of the
numerical contract (see kge_common.h): fused multiply-adds are explicit fmaf().
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['score_triples.hip', 'lp_prep.hip', 'lp_gemm_mfma.hip', 'lp_split_mfma.hip', 'lp_hi_stream.hip', 'lp_hi_chunk.hip', 'lp_direct.hip',
           'lp_l1_sad.hip', 'rank_filter.hip', 'corrupt.hip', 'key_sort.hip', 'index_build.hip']
HEADERS = ['kge_common.h', os.path.join('..', '..', 'include', 'kge_hip.h')]
LIB = os.path.join(HERE, 'libkge_hip.so')
# the RCCL exchange step of the sharded path (include/kge_hip_coll.h): its own shared object, so that
# libkge_hip.so does not depend on librccl
COLL_SRC, COLL_LIB = 'collectives.hip', os.path.join(HERE, 'libkge_hip_coll.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-Wall', '-Wno-unused-function']
# per-file extras.  lp_direct.hip: the SLP vectoriser turns the L1 inner loop (sub, then add |.|) into v_pk_add_f32
# pairs -- which issue at HALF rate on gfx950 (tools/probe/valu_rate_probe.hip: 34 T vs 63 T lane-ops/s) and have no
# abs modifier, so every element pays an extra v_and: 3 issue slots per element instead of 2.
# The MFMA count kernels (lp_hi_stream.hip, lp_hi_chunk.hip, lp_split_mfma.hip): NO SLP-packed f32 arithmetic.  r05 found
# ~1 pair in 3e6 of the TransH / TransD epilogue of the free-running kernel miscounted, differently from launch to launch;
# r06 bisected it at INSTRUCTION level (tools/probe/asm_patch_build.py + slp_bisect.sh, profiles/r06/slp_bisect.txt): the only
# instructions that matter are  v_pk_fma_f32 D, A, B, C op_sel:[0,1,0]  -- the LOW result lane taking the HIGH dword of a
# VGPR-pair source.  Rewriting just those (12 per kernel) as two v_fma_f32, as the plain packed form on materialised pairs, or
# recomputing only their low lane removes every mismatch; the same instructions in two register copies, behind 1000 cycles
# of s_nop, or with only src2's op_sel_hi:[.,.,0] (high lane from the low dword) do not / are fine.  In isolation the
# instruction is correct (tools/probe/pk_opsel_probe.hip: 0 wrong of 1e10 beside MFMA / VALU / VMEM traffic) -- it misreads
# only inside the kernel, where the partner wave's MFMAs co-execute with this wave's epilogue: not a missing wait state
# (more nops -> MORE errors), not a source race.  The vectoriser emits that form whenever two scalars it wants to splat sit
# in one register pair; without it no packed f32 instruction of these files selects a high dword for the low lane
# (tests/test_host_logic.py::test_mfma_kernels_hold_no_lane_crossing_packed_f32_operand checks the built ISA).  The sources
# refuse to compile without the flag (KGE_BUILD_NO_SLP).
_NO_SLP = ['-fno-slp-vectorize', '-DKGE_BUILD_NO_SLP=1']
EXTRA_FLAGS = {'lp_direct.hip': ['-fno-slp-vectorize'], 'lp_hi_stream.hip': _NO_SLP, 'lp_hi_chunk.hip': _NO_SLP,
               'lp_split_mfma.hip': _NO_SLP}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def rocm_roots():
    """ROCm installation prefixes to look under: $ROCM_PATH / $ROCM_HOME first, then hipcc's own prefix, then /opt/rocm."""
    roots = [os.environ.get(k) for k in ('ROCM_PATH', 'ROCM_HOME')]
    hipcc = shutil.which('hipcc')
    if hipcc:
        roots.append(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))))
    roots.append('/opt/rocm')
    out = []
    for r in roots:
        if r and r not in out:
            out.append(r)
    return out


def rccl_install():
    """(include dir, library dir) of the RCCL to link libkge_hip_coll.so against, or None when there is none.

    ONE definition for build() here and __graft_entry__.build(): the header AND the linker name `librccl.so` (what `-lrccl`
    resolves; a bare librccl.so.1 run-time library cannot be linked against) under the same prefix."""
    for root in rocm_roots():
        for inc in (os.path.join(root, 'include'),):
            if not (os.path.exists(os.path.join(inc, 'rccl', 'rccl.h')) or os.path.exists(os.path.join(inc, 'rccl.h'))):
                continue
            for libdir in (os.path.join(root, 'lib'), os.path.join(root, 'lib64')):
                if os.path.exists(os.path.join(libdir, 'librccl.so')):
                    return inc, libdir
    return None


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libkge_hip.so.  Returns the path."""
    bdir = os.path.join(HERE, '_build')
    os.makedirs(bdir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(bdir, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs + [os.path.abspath(__file__)]):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + os.environ.get('KGE_HIPCC_EXTRA', '').split() +
                        ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), r.stdout))
        return r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            outs = list(ex.map(run, jobs))
        if verbose:
            for o in outs:
                if o.strip():
                    print(o)
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    cs = os.path.join(HERE, COLL_SRC)
    if force or _stale(COLL_LIB, [cs, os.path.join(HERE, '..', '..', 'include', 'kge_hip_coll.h')] + hdrs):
        # (-lrccl resolves to whichever librccl.so.1 the process has loaded first -- torch's own when the host is Python)
        # Optional ONLY where RCCL itself is missing (rccl_install() is None): such a box still gets the single-GPU core
        # (libkge_hip.so) and the sharded path through the C-ABI (torchkge_amd/_hip_coll.py) fails loudly on first use.
        # Where RCCL is installed a compile error in collectives.hip is an error (run() raises) -- "missing" and "does
        # not compile" are never the same outcome.
        rccl = rccl_install()
        if rccl is not None:
            inc, libdir = rccl
            run([hipcc] + FLAGS + ['-I' + inc, '-shared', '-o', COLL_LIB, cs, '-L' + libdir, '-lrccl', '-Wl,-rpath,' + libdir])
        else:
            import warnings
            if os.path.exists(COLL_LIB):
                os.remove(COLL_LIB)
            warnings.warn('torchkge_amd: libkge_hip_coll.so not built: rccl.h + librccl.so not found under %s'
                          % ', '.join(rocm_roots()))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
