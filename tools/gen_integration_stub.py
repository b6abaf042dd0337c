#!/usr/bin/env python3
"""include/durf_hip.h -> the reference-side ctypes binding (include/durf_ctypes_stub.py) and the copy of it
embedded in INTEGRATION.md, and the product's own table durf_amd/_sigs.py, so that no copy can drift from the header:
    python tools/gen_integration_stub.py            # rewrite both
    python tools/gen_integration_stub.py --check    # exit 1 if either is stale (tests/test_host_logic.py)
The parser handles exactly the C subset the header uses (scalar / pointer parameters, /* */ comments)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, 'include', 'durf_hip.h')
STUB = os.path.join(ROOT, 'include', 'durf_ctypes_stub.py')
SIGS = os.path.join(ROOT, 'durf_amd', '_sigs.py')          # the product's own binding table (durf_amd/_lib.py)
DOC = os.path.join(ROOT, 'INTEGRATION.md')
BEGIN, END = '<!-- BEGIN GENERATED: tools/gen_integration_stub.py -->', '<!-- END GENERATED -->'

SCALARS = {'int': 'i32', 'float': 'f32', 'size_t': 'u64', 'int32_t': 'i32', 'uint32_t': 'C.c_uint32'}


def ctype_of(param):
    """C parameter declaration -> ctypes spelling used in the stub"""
    p = re.sub(r'\bconst\b', '', param).strip()
    name = re.search(r'(\w+)\s*$', p).group(1)
    t = p[:p.rfind(name)].strip()
    stars = t.count('*')
    base = t.replace('*', '').strip()
    if stars == 0:
        return SCALARS[base], name
    if stars == 1 and base == 'float' and name in ('barf_w', 'mults', 'cams_host'):
        return 'C.POINTER(f32)', name          # HOST float arrays (documented as such in the header)
    if stars == 1 and base == 'size_t':
        return 'C.POINTER(u64)', name          # HOST array (per-segment row capacities)
    if stars == 1 and base == 'int':
        return 'C.POINTER(i32)', name          # HOST array (per-segment rows per ray)
    if stars == 2:
        return 'C.POINTER(vp)', name           # host array of device pointers
    return 'vp', name                          # device pointer / stream


def parse_header(text=None):
    """-> [(name, restype, [(ctype, param name)...])] in header order"""
    text = open(HDR).read() if text is None else text
    body = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    out = []
    for m in re.finditer(r'(?:^|\n)\s*(const char\*|int|size_t)\s+(durf_\w+)\s*\(([^;{]*?)\)\s*;', body):
        ret, name, args = m.group(1), m.group(2), ' '.join(m.group(3).split())
        params = [] if args in ('', 'void') else [ctype_of(a) for a in args.split(',')]
        out.append((name, {'const char*': 'C.c_char_p', 'int': 'i32', 'size_t': 'u64'}[ret], params))
    return out


def render():
    fns = parse_header()
    L = ['"""ctypes binding of libdurf_hip.so for a reference-side caller -- GENERATED from include/durf_hip.h by',
         'tools/gen_integration_stub.py (do not edit; `--check` runs in the CPU test-suite).  All `vp` arguments are',
         'DEVICE pointers of caller-owned buffers except `stream` (hipStream_t); C.POINTER(...) arguments are host',
         'arrays.  Every call is asynchronous on `stream` and returns 0 or an error code (durf_last_error())."""',
         'import ctypes as C', '', 'vp, i32, f32, u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t', '', '',
         'def bind(path=\'durf_amd/libdurf_hip.so\'):', '    L = C.CDLL(path)']
    for name, ret, params in fns:
        L.append('    L.%s.restype = %s' % (name, ret))
        args = ', '.join(t for t, _ in params)
        names = ', '.join(n for _, n in params)
        line = '    L.%s.argtypes = [%s]' % (name, args)
        L.append(line)
        if names:
            L.append('    #   (%s)' % names)
    L += ['    return L', '']
    return '\n'.join(L)


def render_sigs():
    L = ['"""Argument types of every entry point of libdurf_hip.so -- GENERATED from include/durf_hip.h by',
         'tools/gen_integration_stub.py (do not edit; `--check` runs in the CPU test-suite)."""', 'import ctypes as C', '',
         'vp, i32, f32, u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t', '', 'SIGS = {']
    for name, ret, params in parse_header():
        L.append('    %r: (%s, [%s]),' % (name, ret, ', '.join(t for t, _ in params)))
    L += ['}', '']
    return '\n'.join(L)


def doc_with_stub(doc, stub):
    a, b = doc.index(BEGIN), doc.index(END)
    return doc[:a + len(BEGIN)] + '\n```python\n' + stub + '```\n' + doc[b:]


def main():
    stub = render()
    sigs = render_sigs()
    doc = open(DOC).read()
    new_doc = doc_with_stub(doc, stub)
    if '--check' in sys.argv:
        ok = (os.path.exists(STUB) and open(STUB).read() == stub and new_doc == doc and
              os.path.exists(SIGS) and open(SIGS).read() == sigs)
        if not ok:
            print('stale: run python tools/gen_integration_stub.py')
        sys.exit(0 if ok else 1)
    open(STUB, 'w').write(stub)
    open(SIGS, 'w').write(sigs)
    open(DOC, 'w').write(new_doc)
    print('wrote %s (%d functions) and the block in INTEGRATION.md' % (os.path.relpath(STUB, ROOT), len(parse_header())))


if __name__ == '__main__':
    main()
