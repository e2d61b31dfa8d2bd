"""Developer / test tool: the kernel descriptors (registers, scratch, LDS, spills) of every gfx950 kernel inside a HIP shared
library, read from the code objects' AMDGPU metadata notes - no external binary needed (pure Python + msgpack).

    python tools/code_objects.py [path/to/lib.so] [name-substring ...]      -> one line per kernel

A HIP .so keeps its device code in the `.hip_fatbin` section: one clang offload bundle per translation unit
("__CLANG_OFFLOAD_BUNDLE__", optionally zlib/zstd-compressed as "CCOB"), each holding one ELF per target.  Every ELF carries a
note (owner "AMDGPU", type 32) whose descriptor is the msgpack-encoded `amdhsa.kernels` list."""
import os
import struct
import sys
import zlib

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(blob):
    assert blob[:4] == b"\x7fELF" and blob[4] == 2, "ELF64 expected"
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    stroff = secs[shstrndx][2]

    def nm(o):
        e = blob.index(b"\0", stroff + o)
        return blob[stroff + o:e].decode()
    return [(nm(n), t, o, s) for (n, t, o, s) in secs]


def _bundles(fat):
    """Yield the (triple, bytes) entries of every bundle in a .hip_fatbin section."""
    pos = 0
    while True:
        a, c = fat.find(MAGIC, pos), fat.find(b"CCOB", pos)
        if c >= 0 and (a < 0 or c < a):                       # compressed bundle: header, then the deflated/zstd payload
            ver, method = struct.unpack_from("<HH", fat, c + 4)
            if ver >= 2:
                total, = struct.unpack_from("<I" if ver == 2 else "<Q", fat, c + 8)
                hdr = 8 + (4 if ver == 2 else 8) + (4 if ver == 2 else 8) + 8
            else:
                total, hdr = None, 8 + 4 + 8
            payload = fat[c + hdr: c + total if total else None]
            if method == 0:
                data = zlib.decompress(payload)
            else:
                raise RuntimeError("zstd-compressed offload bundle: build with -no-offload-compress")
            yield from _bundles(data)
            pos = c + (total or len(fat))
            continue
        if a < 0:
            return
        n, = struct.unpack_from("<Q", fat, a + len(MAGIC))
        p = a + len(MAGIC) + 8
        end = a + len(MAGIC)
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24: p + 24 + tl].decode()
            p += 24 + tl
            yield triple, fat[a + off: a + off + size]
            end = max(end, a + off + size)
        pos = end


def _notes(elf):
    for name, typ, off, size in _elf_sections(elf):
        if typ != 7:                                          # SHT_NOTE
            continue
        p, e = off, off + size
        while p + 12 <= e:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            owner = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if owner == b"AMDGPU" and ntype == 32:
                yield desc


def kernels(path, target="gfx950"):
    """[{name, vgpr, agpr, sgpr, scratch, lds, vgpr_spill, sgpr_spill, max_threads}] for every kernel of the library."""
    import msgpack
    blob = open(path, "rb").read()
    fat = b"".join(blob[o:o + s] for (n, t, o, s) in _elf_sections(blob) if n == ".hip_fatbin")
    out = []
    for triple, code in _bundles(fat):
        if target not in triple or code[:4] != b"\x7fELF":
            continue
        for desc in _notes(code):
            md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in md.get("amdhsa.kernels", []):
                out.append({"name": k.get(".name", ""), "vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0),
                            "sgpr": k.get(".sgpr_count", 0), "scratch": k.get(".private_segment_fixed_size", 0),
                            "lds": k.get(".group_segment_fixed_size", 0), "vgpr_spill": k.get(".vgpr_spill_count", 0),
                            "sgpr_spill": k.get(".sgpr_spill_count", 0), "max_threads": k.get(".max_flat_workgroup_size", 0)})
    return out


def demangle(names):
    import shutil
    import subprocess
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return out[:len(names)]
    except Exception:
        return list(names)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(root, "ssr_eval_amd", "libssrhip.so")
    ks = kernels(lib)
    for k, d in zip(ks, demangle([k["name"] for k in ks])):
        k["demangled"] = d
    ks = [k for k in ks if not args or any(a in k["demangled"] for a in args)]
    for k in sorted(ks, key=lambda k: k["demangled"]):
        print("%-110s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d spill v%d s%d" % (
            k["demangled"][:110], k["vgpr"], k["agpr"], k["sgpr"], k["scratch"], k["lds"], k["vgpr_spill"], k["sgpr_spill"]))
    print("%d kernels; with scratch: %d; with sgpr spills: %d" % (
        len(ks), sum(1 for k in ks if k["scratch"]), sum(1 for k in ks if k["sgpr_spill"])))
