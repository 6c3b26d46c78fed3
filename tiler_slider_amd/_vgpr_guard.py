"""Build-time guard against the gfx950 "last VGPR" hazard (profiles/r03_wrong_slide_isa.md).

On gfx950 a 64-bit shift (v_lshlrev_b64) whose 32-bit shift-amount operand is the HIGHEST VGPR of the wave's register
allocation (v31 of a 32-register kernel) occasionally reads the wave's v0 instead when a second wave shares the SIMD: tiles
slid past their row in 2-7 % of 8x8 boards.  Neither the compiler's register allocator nor its hazard recognizer knows about
it.  The allocation granule is 8 VGPRs, so the last ALLOCATED register is one the code touches only when the kernel's VGPR
count is a multiple of 8.  Two tools, both used by _cabi.build_library:

  scan_code_object  disassembles a gfx950 code object and lists, per kernel, every VALU instruction with a 64-bit operand
                    that reads the last register of the kernel's allocation:
                      class A  the observed pattern: v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 with it as shift amount
                      class B  any other such instruction that reads it, alone or as the upper half of a register pair
  pad_vgpr_allocations  gives a kernel one register more in its descriptor (allocation + 8), so that the last allocated
                    register is never one the code touches.

Policy (round 4): below 64 registers a granule more costs no occupancy (8 waves per SIMD either way) - every kernel whose
count fills its allocation is padded, hit or not.  From 64 on a granule costs a wave per SIMD (64 -> 72: 8 -> 7 waves, 96 ->
104: 5 -> 4, 128 -> 136: 4 -> 3), so those are padded only when the scanner finds a class A / B read in the UNPADDED object.
The final object is scanned again and the build fails on any hit.
"""
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
GRANULE = 8        # gfx90a+ allocate VGPRs in blocks of 8
SIMD_VGPRS = 512   # per lane and SIMD: waves per SIMD = min(8, 512 // allocation)
FREE_BELOW = 64    # allocations up to here keep 8 waves per SIMD with one granule more

_SHIFT64 = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")
_WIDE = re.compile(r"^v_\w*(b64|u64|i64|f64)\w*$")
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def waves_per_simd(vgprs):
    alloc = -(-max(vgprs, 1) // GRANULE) * GRANULE
    return min(8, SIMD_VGPRS // alloc)


def unbundle(lib, workdir):
    """The gfx950 code object inside a built shared library."""
    fat, co = os.path.join(workdir, "fat.bin"), os.path.join(workdir, "dev.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(workdir, "copy.so")], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    return co


def vgpr_counts(notes):
    """kernel symbol -> (vgpr_count, agpr_count) from the code object's metadata note"""
    out, name, agpr = {}, None, 0
    for line in notes.splitlines():
        s = line.strip()
        if s.startswith("- .agpr_count:") or s.startswith(".agpr_count:"):
            agpr = int(s.split(":")[1])
        elif s.startswith(".name:"):
            name = s.split(":", 1)[1].strip()
        elif s.startswith(".vgpr_count:") and name:
            out[name] = (int(s.split(":")[1]), agpr)
            name = None
    return out


def scan_code_object(co):
    """Returns (findings_A, findings_B, counts): lists of (kernel, instruction text); counts = kernel -> (vgprs, agprs)."""
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    counts = vgpr_counts(notes)
    a, b, kernel, last = [], [], None, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel = m.group(1)
            last = None
            if kernel in counts:
                n, ag = counts[kernel]
                total = n + ag
                if total and total % GRANULE == 0 and ag == 0:  # the last allocated register is in use only then
                    last = total - 1
            continue
        if last is None:
            continue
        text = line.split("//")[0].strip()
        if not text.startswith("v_"):
            continue
        mnem, _, ops = text.partition(" ")
        mnem_base = mnem.replace("_e32", "").replace("_e64", "")
        if not _WIDE.match(mnem_base):
            continue
        operands = [o.strip() for o in ops.split(",")]
        srcs = operands[1:]  # operand 0 is the destination (v_cmp writes an SGPR pair / vcc: also not a VGPR source)
        hit_scalar = hit_pair = False
        for k, o in enumerate(srcs):
            for r1, lo, hi in _REG.findall(o):
                if r1 and int(r1) == last:
                    hit_scalar = True
                    if mnem_base in _SHIFT64 and k == 0:
                        a.append((kernel, text))
                elif hi and int(hi) == last:
                    hit_pair = True
        if (hit_scalar or hit_pair) and not (a and a[-1] == (kernel, text)):
            b.append((kernel, text))
    return a, b, counts


def scan(lib):
    """The same for a built shared library: (findings_A, findings_B, number of kernels)."""
    with tempfile.TemporaryDirectory() as wd:
        a, b, counts = scan_code_object(unbundle(lib, wd))
    return a, b, len(counts)


def pad_vgpr_allocations(asm, hits=None):
    """Device assembly -> (patched assembly, {kernel: (vgprs before, reason)}).  A kernel whose VGPR count fills its
    allocation exactly gets one register more in its descriptor and metadata when that costs no wave per SIMD - below
    FREE_BELOW, and at 88 / 104 / 112 / 120 registers, where the next granule holds as many waves ("free") -, or when `hits`
    (kernel names with a scanner finding in the unpadded object) names it ("hit").
    hits=None pads every such kernel (round 3's blanket rule)."""
    padded, out, in_meta, name, kernel = {}, [], False, None, None
    for line in asm.split("\n"):
        t = line.strip()
        if t.startswith(".amdhsa_kernel "):
            kernel = t.split()[1]
        elif t.startswith(".amdhsa_next_free_vgpr ") and kernel:
            n = int(t.split()[1])
            if n > 0 and n % GRANULE == 0:
                free = n < FREE_BELOW or waves_per_simd(n + 1) == waves_per_simd(n)  # (88, 104, 112, 120: the next granule holds as many waves)
                reason = "free" if free else "hit" if (hits is None or kernel in hits) else None
                if reason:
                    if n + 1 > SIMD_VGPRS:
                        raise ValueError(f"{kernel}: cannot pad {n} VGPRs")
                    line = line.replace(str(n), str(n + 1))
                    padded[kernel] = (n, reason)
        elif t == "amdhsa.kernels:":
            in_meta = True
        elif in_meta and t.startswith(".name:"):
            name = t.split(":", 1)[1].strip()
        elif in_meta and t.startswith(".vgpr_count:") and name in padded:
            n = int(t.split(":")[1])
            line = re.sub(r"\d+\s*$", str(n + 1), line)
        out.append(line)
    return "\n".join(out), padded
