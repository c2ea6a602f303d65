"""bindings/rust/src/gpu.rs is uncompiled source (no Rust toolchain in this image): at least keep it in step
with the header -- every symbol the header declares is bound, nothing else is, and argument counts agree."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_decls():
    text = open(os.path.join(ROOT, "include", "soundsym_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"SSYM_API\s+[\w\s\*]+?\b(ssym_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _rust_decls():
    text = open(os.path.join(ROOT, "bindings", "rust", "src", "gpu.rs")).read()
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    block = re.sub(r"/\*.*?\*/", "", block, flags=re.S)
    block = re.sub(r"//[^\n]*", "", block)
    out = {}
    for m in re.finditer(r"pub fn (ssym_\w+)\s*\(([^;]*?)\)\s*->", block, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_rust_binding_covers_the_header():
    c, r = _c_decls(), _rust_decls()
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, n in c.items():
        assert r[name] == n, (name, n, r[name])


def test_rust_struct_fields_follow_the_header():
    import soundsym_amd._native as nat
    text = open(os.path.join(ROOT, "bindings", "rust", "src", "gpu.rs")).read()
    for struct, cls in (("SsymConfig", nat.Config), ("SsymTimings", nat.Timings)):
        body = text[text.index("pub struct %s {" % struct):]
        body = body[:body.index("}")]
        fields = re.findall(r"pub (\w+):", body)
        assert fields == [f for f, _ in cls._fields_], struct
