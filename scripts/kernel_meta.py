"""VGPRs / spills / scratch / LDS of the kernels of a built object or library: python scripts/kernel_meta.py <file.o|.so> [filter]
(llvm-objdump --offloading extracts the gfx950 code object; llvm-readelf --notes prints the kernel descriptors' metadata)."""
import glob, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
src, flt = os.path.abspath(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as d:
    tmp = os.path.join(d, os.path.basename(src))
    os.symlink(src, tmp)
    subprocess.run([LLVM + "llvm-objdump", "--offloading", tmp], capture_output=True, cwd=d)
    cos = [f for f in glob.glob(tmp + ".*") if "amdgcn" in f]
    for co in cos:
        txt = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [0, "?"])[1]
            name = g("name")
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            if flt not in name:
                continue
            print(name.replace("igx::", "")[:120], "| vgpr", g("vgpr_count"), "spill", g("vgpr_spill_count"), "sgpr_spill", g("sgpr_spill_count"), "scratch", g("private_segment_fixed_size"), "lds", g("group_segment_fixed_size"))
