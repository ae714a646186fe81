#!/usr/bin/env python3
"""Developer probe: per-kernel register / spill / scratch / LDS figures of the built library, read from the code-object
metadata (.hip_fatbin -> bundles -> gfx950 code objects -> AMDGPU metadata note)."""
import os, re, subprocess, sys, tempfile

B = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "dex-nerf_amd", "lib", "libdexnerf_hip.so")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory(dir=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")) as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([f"{B}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    offs = [m.start() for m in re.finditer(re.escape(magic), blob)]
    rows = []
    for n, o in enumerate(offs):
        bf = os.path.join(d, f"b{n}")
        open(bf, "wb").write(blob[o:(offs[n + 1] if n + 1 < len(offs) else len(blob))])
        co = bf + ".co"
        subprocess.call([f"{B}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                         f"--input={bf}", f"--output={co}"], stderr=subprocess.DEVNULL)
        if not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.check_output([f"{B}/llvm-readelf", "--notes", co], text=True)
        for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            blk = ".agpr_count:" + blk
            get = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
            name = subprocess.check_output(["c++filt", get("name")], text=True).strip()
            rows.append((name, get("vgpr_count"), get("agpr_count"), get("sgpr_count"), get("vgpr_spill_count") + "/" + get("sgpr_spill_count"),
                         get("private_segment_fixed_size"), get("group_segment_fixed_size"), get("max_flat_workgroup_size")))
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spill':>5} {'scratch':>7} {'lds':>6} {'wg':>5}  kernel")
    for r in rows:
        if pat in r[0]:
            print(f"{r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>5} {r[5]:>7} {r[6]:>6} {r[7]:>5}  {r[0][:150]}")
