"""Mechanical check of nim/laser_hip.nim against the C-ABI -- the strongest check available without a Nim compiler
(none in this image).  Every `importc` is parsed out of the shim and compared with `nm -D liblaser_hip.so` and with the
preprocessed include/laser_hip.h: symbol exists, same number of parameters, C-compatible parameter and return types.
Also: the exported Nim procs keep the reference's names and signatures for this path (gemm.nim:184-193,
gemm_prepacked.nim:76-292, swapaxes.nim:16-112, conv2d_im2col.nim:10-100, blas.nim:18-23)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NIM = os.path.join(ROOT, "nim", "laser_hip.nim")
SO = os.path.join(ROOT, "laser_amd", "lib", "liblaser_hip.so")

NIM_CLASS = {"int": "i64", "Natural": "i64", "cint": "i32", "int32": "i32", "int64": "i64", "float32": "f32",
             "float64": "f64", "pointer": "ptr", "cstring": "ptr"}


def nim_class(t):
    t = t.strip()
    if t.startswith("ptr ") or t.startswith("ptr("):
        return "ptr"
    return NIM_CLASS[t]


def c_class(t):
    t = t.strip()
    if "*" in t:
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).strip()
    base = t.split()[0] if len(t.split()) > 1 and t.split()[-1].isidentifier() and t.split()[0] in (
        "int64_t", "int32_t", "int", "float", "double", "long") else t
    base = base.split()[0]
    return {"int64_t": "i64", "int32_t": "i32", "int": "i32", "float": "f32", "double": "f64", "void": "void"}[base]


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def nim_params(plist):
    """'M, N, K: int, alpha: float32, A: ptr float32' -> ['i64','i64','i64','f32','ptr']"""
    classes, pending = [], 0
    for group in split_top(plist, ","):
        if ":" in group:
            names, typ = group.split(":", 1)
            pending += 1
            classes += [nim_class(typ)] * pending
            pending = 0
        else:
            pending += 1
    assert pending == 0, plist
    return classes


def parse_nim_imports():
    src = open(NIM).read()
    pat = re.compile(r'^proc\s+(\w+)\*?\s*\((.*)\)\s*(?::\s*([\w ]+?))?\s*\{\.\s*lh\s*,\s*importc:\s*"(\w+)"\s*\.\}', re.M)
    imports = []
    for m in pat.finditer(src):
        name, plist, ret, sym = m.groups()
        imports.append({"nim": name, "params": nim_params(plist) if plist.strip() else [],
                        "ret": nim_class(ret) if ret else "void", "sym": sym})
    return src, imports


def header_protos():
    src = subprocess.run(["gcc", "-E", "-P", os.path.join(ROOT, "include", "laser_hip.h")], check=True,
                         capture_output=True, text=True).stdout
    src = re.sub(r"\s+", " ", src)
    protos = {}
    for m in re.finditer(r"([\w ]+?[\w\*]) \*?(laser_hip_\w+) ?\(([^)]*)\) ?;", src):
        ret, name, plist = m.groups()
        ret = ret.strip()
        if "*" in m.group(0).split(name)[0]:
            ret = ret + " *"
        params = [] if plist.strip() in ("", "void") else [c_class(p) for p in plist.split(",")]
        protos[name] = {"ret": c_class(ret), "params": params}
    return protos


@pytest.fixture(scope="module")
def exported():
    if not os.path.exists(SO):
        import __graft_entry__ as g
        g.build()
    out = subprocess.run(["nm", "-D", "--defined-only", SO], check=True, capture_output=True, text=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_every_importc_binds_a_real_symbol_with_matching_signature(exported):
    src, imports = parse_nim_imports()
    protos = header_protos()
    assert len(imports) >= 50, f"only {len(imports)} importc declarations parsed -- parser or shim broken"
    # every importc in the file was parsed (none hidden in a form the parser does not understand)
    code = "\n".join(line.split("#", 1)[0] for line in src.splitlines())
    assert code.count("importc") == len(imports), \
        "an importc is spelled in a way this check does not parse (must be: proc ...{.lh, importc: \"sym\".})"
    seen = set()
    for imp in imports:
        sym = imp["sym"]
        assert imp["nim"] == sym, f"Nim-side name {imp['nim']} differs from its C symbol {sym}"
        assert sym not in seen, f"{sym} imported twice"
        seen.add(sym)
        assert sym in exported, f"{sym}: not exported by liblaser_hip.so"
        assert sym in protos, f"{sym}: not declared in include/laser_hip.h"
        c = protos[sym]
        assert len(imp["params"]) == len(c["params"]), f"{sym}: {len(imp['params'])} Nim parameters vs {len(c['params'])} in C"
        assert imp["params"] == c["params"], f"{sym}: parameter types differ\n nim {imp['params']}\n c   {c['params']}"
        assert imp["ret"] == c["ret"], f"{sym}: return type {imp['ret']} (Nim) vs {c['ret']} (C)"


def test_no_identifier_pasting_and_no_underscore_identifiers():
    src, _ = parse_nim_imports()
    code = "\n".join(line.split("#", 1)[0] for line in src.splitlines())
    assert "`" not in code, "backtick identifier construction: spell every import out with importc: \"...\""
    for tok in re.findall(r"(?<![\w\"])_\w+", code):
        raise AssertionError(f"Nim identifiers cannot start with an underscore: {tok}")
    assert re.search(r"importc\s*\.\}", code) is None and re.search(r"importc\s*,", code) is None, \
        "bare `importc` without a symbol string"


def test_reference_signatures_are_kept():
    src, _ = parse_nim_imports()
    flat = re.sub(r"\s+", " ", re.sub(r"#.*", "", src))
    # gemm.nim:184-193
    assert ("proc gemm_strided*[T: SomeNumber]( M, N, K: int, alpha: T, A: ptr T, rowStrideA, colStrideA: int, "
            "B: ptr T, rowStrideB, colStrideB: int, beta: T, C: ptr T, rowStrideC, colStrideC: int)") in flat
    assert "elif T is uint32:" in flat            # gemm.nim:239 dispatches int32 and uint32 together
    # gemm_prepacked.nim:76-85, 111-135, 157-218, 275-292
    assert "proc gemm_prepackB_mem_required*(T: type, M, N, K: int): int" in flat
    assert "proc gemm_prepackA_mem_required*(T: type, M, N, K: int): int" in flat
    assert "proc gemm_prepackB*[T](dst_packedB: ptr (T or UncheckedArray[T]), M, N, K: int, src_B: ptr T, rowStrideB, colStrideB: int)" in flat
    assert "proc gemm_prepackA*[T](dst_packedA: ptr (T or UncheckedArray[T]), M, N, K: int, src_A: ptr T, rowStrideA, colStrideA: int)" in flat
    assert "proc gemm_packed*[T: SomeNumber](M, N, K: int, alpha: T, packedA: ptr (T or UncheckedArray[T]), packedB: ptr (T or UncheckedArray[T]), beta: T, C: ptr (T or UncheckedArray[T]), rowStrideC, colStrideC: int)" in flat
    # swapaxes.nim:16-112
    assert "proc transpose2D_copy*[T](dst, src: ptr (T or UncheckedArray[T]), NR, NC: Natural)" in flat
    assert "proc transpose2D_batched*[T](dst, src: ptr (T or UncheckedArray[T]), N, NR, NC: Natural)" in flat
    assert "proc nchw2nhwc*[T](dst_nhwc, src_nchw: ptr (T or UncheckedArray[T]), N, C, H, W: Natural)" in flat
    assert "proc nhwc2nchw*[T](dst_nchw, src_nhwc: ptr (T or UncheckedArray[T]), N, C, H, W: Natural)" in flat
    # conv2d_im2col.nim:10-20, 22-31, 90-100 (Tensor[float32], which conv2d_common.nim:13 defines as seq[float32])
    assert "proc im2col_workspace_size*(ishape: TensorShape, kshape: KernelShape, padding: Padding, strides: Strides): int" in flat
    assert ("proc conv2d_im2col*( output: var Tensor[float32], oshape: TensorShape, input: Tensor[float32], "
            "ishape: TensorShape, kernel: Tensor[float32], kshape: KernelShape, padding: Padding, strides: Strides, "
            "pworkspace: ptr float32 )") in flat
    assert "Tensor*[T] = seq[T]" in flat
    # blas.nim:18-23: both overloads
    for ft in ("float32", "float64"):
        assert (f"proc gemm*(ORDER: OrderType, TRANSA, TRANSB: TransposeType, M, N, K: int, ALPHA: {ft}, A: ptr {ft}, "
                f"LDA: int, B: ptr {ft}, LDB: int, BETA: {ft}, C: ptr {ft}, LDC: int)") in flat


def test_device_pointers_are_a_distinct_type_without_unsafe_raw_data():
    """The guard of VERDICT r1 #9: a HipStorage tensor cannot reach Laser's host forEach, whose contract needs
    `unsafe_raw_data` (foreach.nim:7-29) -- the device accessor has another name and a distinct, non-indexable type."""
    src, _ = parse_nim_imports()
    code = "\n".join(line.split("#", 1)[0] for line in src.splitlines())
    assert re.search(r"DevicePtr\*\[T\]\s*=\s*distinct pointer", code)
    assert re.search(r"raw_buffer\*:\s*DevicePtr\[T\]", code)
    assert "unsafe_device_data*" in code
    assert not re.search(r"(func|proc|template)\s+unsafe_raw_data", code), "the shim must not give device storage an unsafe_raw_data"
    assert not re.search(r"(func|proc|template)\s+`\[\]`", code), "DevicePtr must stay non-indexable on the host"
