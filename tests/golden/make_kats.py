#!/usr/bin/env python3
"""Extract the reference's known-answer tests for the GEMM / conv hot path into JSON.

Runs ONLY in the build container (needs /root/reference); the output
tests/golden/laser_kats.json is committed so tests never read the reference at run time.

Sources (file:line, relative to /root/reference):
  laser/primitives/matrix_multiplication/gemm.nim:255-507            8 GEMM KATs
  laser/primitives/matrix_multiplication/gemm_prepacked.nim:352-523  9 pre-packed GEMM KATs
  benchmarks/convolution/conv2d_common.nim:139-283                   2 conv KATs
All are alpha=1, beta=0, row-major contiguous, exact-equality expectations on small-integer
data, so they are valid for f32/f64/i32/i64 alike.
"""
import json, re, sys, os

REF = "/root/reference"


def balanced(src, start):
    """src[start] == '[' -> return index one past its matching ']'."""
    depth = 0
    for i in range(start, len(src)):
        if src[i] == "[":
            depth += 1
        elif src[i] == "]":
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced")


def nim_literal(text):
    text = re.sub(r"#[^\n]*", "", text)  # comments
    text = re.sub(r"\bfloat32\b|\bfloat64\b|\bint32\b|\bint\b", "", text)
    text = re.sub(r"'[fi]\d+", "", text)  # literal suffixes
    return eval(text, {"__builtins__": {}})


def find_lets(src, names):
    out = []
    for m in re.finditer(r"let\s+(\w+)(?:\{\.inject\.\})?\s*=\s*\[", src):
        if m.group(1) in names:
            s = m.end() - 1
            e = balanced(src, s)
            out.append((m.group(1), nim_literal(src[s:e]), src.count("\n", 0, m.start()) + 1))
    return out


def gemm_kats(path, label):
    src = open(os.path.join(REF, path)).read()
    src = src[src.index("when isMainModule"):]
    base_line = open(os.path.join(REF, path)).read().count("\n", 0, open(os.path.join(REF, path)).read().index("when isMainModule"))
    lets = find_lets(src, {"a", "b", "ab"})
    kats = []
    i = 0
    while i + 2 < len(lets) + 0 and i + 2 <= len(lets) - 1:
        (na, a, la), (nb, b, _), (nab, ab, _) = lets[i], lets[i + 1], lets[i + 2]
        assert (na, nb, nab) == ("a", "b", "ab"), (na, nb, nab)
        M, K, N = len(a), len(a[0]), len(b[0])
        assert len(b) == K and len(ab) == M and len(ab[0]) == N
        kats.append({"source": f"{path}:{base_line + la}", "api": label, "M": M, "N": N, "K": K,
                     "A": a, "B": b, "C": ab})
        i += 3
    return kats


def conv_kats(path):
    src = open(os.path.join(REF, path)).read()
    start = src.index("template conv_impl_check")
    base_line = src.count("\n", 0, start)
    body = src[start:]
    lets = find_lets(body, {"input", "kernel", "target"})
    shapes = re.findall(r"ishape\{\.inject\.\}(?:: TensorShape)? = \(([^)]*)\)", body)
    kshapes = re.findall(r"kshape\{\.inject\.\}(?:: KernelShape)? = \(([^)]*)\)", body)
    pads = re.findall(r"padding\{\.inject\.\} = \(([^)]*)\)", body)
    strs = re.findall(r"strides\{\.inject\.\} = \(([^)]*)\)", body)
    kats = []

    def flat(x):
        return [v for s in x for v in flat(s)] if isinstance(x, list) else [x]

    for i in range(len(lets) // 3):
        (ni, inp, li), (nk, ker, _), (nt, tgt, _) = lets[3 * i: 3 * i + 3]
        assert (ni, nk, nt) == ("input", "kernel", "target")
        kats.append({"source": f"{path}:{base_line + li}",
                     "ishape": [int(v) for v in shapes[i].split(",")],
                     "kshape": [int(v) for v in kshapes[i].split(",")],
                     "padding": [int(v) for v in pads[i].split(",")],
                     "strides": [int(v) for v in strs[i].split(",")],
                     "input": flat(inp), "kernel": flat(ker), "target": flat(tgt)})
    return kats


def main():
    out = {
        "generator": "tests/golden/make_kats.py (run in the build container against /root/reference)",
        "gemm": gemm_kats("laser/primitives/matrix_multiplication/gemm.nim", "gemm_strided"),
        "gemm_prepacked": gemm_kats("laser/primitives/matrix_multiplication/gemm_prepacked.nim", "gemm_packed"),
        "conv": conv_kats("benchmarks/convolution/conv2d_common.nim"),
    }
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "laser_kats.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(f"{len(out['gemm'])} gemm, {len(out['gemm_prepacked'])} prepacked, {len(out['conv'])} conv KATs -> {dst}")


if __name__ == "__main__":
    main()
