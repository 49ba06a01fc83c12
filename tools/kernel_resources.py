#!/usr/bin/env python
"""Register / scratch / LDS budget of every gfx950 kernel in a built library, read from the code objects' metadata notes:
    python tools/kernel_resources.py [stabstitch2_amd/libstabstitch_hip.so] [name-substring]
The .so carries one clang offload bundle per translation unit in its .hip_fatbin section; each bundle's gfx950 entry is an ELF
whose NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count / .agpr_count / .sgpr_count / .vgpr_spill_count / .sgpr_spill_count /
.private_segment_fixed_size (scratch bytes per lane) / .group_segment_fixed_size (static LDS).  tests/test_host_logic.py holds the
MFMA kernels to zero spills and zero scratch with `kernels()` below."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(lib_path):
    """-> list of bytes: the gfx950 ELF of every bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, 'fatbin')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, lib_path, os.path.join(d, 'x')],
                       check=True, capture_output=True)
        blob = open(fat, 'rb').read()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from('<Q', blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if 'gfx950' in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


_FIELDS = ('.vgpr_count', '.agpr_count', '.sgpr_count', '.vgpr_spill_count', '.sgpr_spill_count', '.private_segment_fixed_size',
           '.group_segment_fixed_size', '.max_flat_workgroup_size')


def kernels(lib_path):
    """-> {kernel name (demangled where llvm-cxxfilt knows it): {field: int}} over every gfx950 code object of the library,
    plus 'uses_mfma': whether the kernel's disassembly contains a v_mfma instruction."""
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for i, elf in enumerate(code_objects(lib_path)):
            f = os.path.join(d, 'co%d.elf' % i)
            open(f, 'wb').write(elf)
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', f], check=True, capture_output=True, text=True).stdout
            cur = None
            entries = []
            for line in notes.splitlines():
                m = re.match(r'^  - (\.[a-z_]+):\s*(.*)$', line)          # first key of a kernel entry
                if m:
                    cur = {}
                    entries.append(cur)
                else:
                    m = re.match(r'^    (\.[a-z_]+):\s*(.*)$', line)      # kernel-level key (argument entries sit deeper)
                if not m or cur is None:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k in _FIELDS:
                    cur[k] = int(v)
                elif k in ('.name', '.symbol'):
                    cur[k] = v.strip("'\"")
            # which kernels contain matrix instructions
            dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', f], check=True, capture_output=True, text=True).stdout
            mfma = {}
            sym = None
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
                if m:
                    sym = m.group(1)
                    mfma.setdefault(sym, 0)
                elif sym is not None and 'v_mfma' in line:
                    mfma[sym] += 1
            for e in entries:
                sym = e.get('.symbol', '')
                base = sym[:-3] if sym.endswith('.kd') else sym
                name = e.get('.name', base)
                e['mfma_instructions'] = mfma.get(base, 0)
                res[name] = e
    names = list(res)
    try:
        dem = subprocess.run([os.path.join(LLVM, 'llvm-cxxfilt')] + names, check=True, capture_output=True, text=True).stdout.splitlines()
        if len(dem) == len(names):
            res = {d_: res[n] for d_, n in zip(dem, names)}
    except Exception:
        pass
    return res


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stabstitch2_amd', 'libstabstitch_hip.so')
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    ks = kernels(lib)
    print('%-100s %5s %5s %5s %6s %8s %7s %5s' % ('kernel', 'vgpr', 'agpr', 'sgpr', 'spill', 'scratch', 'lds', 'mfma'))
    for name in sorted(ks):
        if pat and pat not in name:
            continue
        e = ks[name]
        print('%-100s %5d %5d %5d %6d %8d %7d %5d' % (name[:100], e.get('.vgpr_count', -1), e.get('.agpr_count', 0), e.get('.sgpr_count', -1),
                                                     e.get('.vgpr_spill_count', 0) + e.get('.sgpr_spill_count', 0),
                                                     e.get('.private_segment_fixed_size', 0), e.get('.group_segment_fixed_size', 0),
                                                     e['mfma_instructions']))


if __name__ == '__main__':
    main()
