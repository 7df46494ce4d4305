"""The step loop of the frame kernel's scatter-ray traversal (trace_queue<1, false>, prt_frame.h / trace_step_phase, prt_device.h)
as gfx950 ISA: compiles prt_kernels.hip to assembly with the product's flags (device side only, no GPU needed) and cuts the loop
out of the listing by the compiler's own loop annotations.

    step_loop_isa.py            -> summary (instruction counts by kind, scratch_ instructions inside the loop: must be none)
    step_loop_isa.py --dump F   -> also writes the annotated loop body to F (profiles/r03_step_loop_isa.txt is one)

The step loop = the innermost loop of that function whose body holds BOTH the record fetches (global_load_dwordx4: four per node, two
and a dwordx3 per triangle) and the lane exchange of the cooperative leaf round (eight ds_bpermute_b32): header label L, every block annotated "in Loop: Header=L" and
the blocks of the loops nested in it.  tests/test_host_cpu.py::test_step_loop_has_no_scratch_access runs this."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prt_amd import _build as B  # noqa: E402

FUNC = "_Z11trace_queueILi1ELb0EEvm"


def device_asm(extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC", "-pthread", "-ldl")]
        cmd = [B.hipcc()] + flags + list(extra) + ["--cuda-device-only", "-S", os.path.join(B.CSRC, "prt_kernels.hip"), "-o", out]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernel_resources(kernel="_Z12frame_kernelILb0ELb0EEv9FrameArgs"):
    """{VGPRs, TotalSGPRs, ScratchSize, LDS Size, Occupancy} of the timed frame kernel, from the compiler's resource-usage remarks."""
    flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC", "-pthread", "-ldl")]
    cmd = [B.hipcc()] + flags + ["--cuda-device-only", "-c", os.path.join(B.CSRC, "prt_kernels.hip"), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, check=True).stderr
    out, on = {}, False
    for ln in err.splitlines():
        if "Function Name:" in ln:
            on = kernel in ln
        elif on:
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
            if m:
                out[m.group(1).strip()] = int(m.group(2))
    return out


def function_body(asm, name=FUNC):
    a = asm.index(f"\n{name}:")
    b = asm.index(f".size\t{name}", a)
    return asm[a:b].splitlines()


def blocks_of(lines):
    """[(label, header comment lines, instruction lines)] in listing order; the text before the first label is block ''."""
    out, label, notes, body = [], "", [], []
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m or re.match(r"^; %bb\.\d+:", ln):
            out.append((label, notes, body))
            label, notes, body = (m.group(1) if m else ln.split(":")[0]), [ln], []
        elif ln.strip().startswith(";") and not body:
            notes.append(ln)
        else:
            body.append(ln)
    out.append((label, notes, body))
    return out


def step_loop(lines):
    blocks = blocks_of(lines)
    # loop membership from the annotations: "in Loop: Header=BBa_b Depth=d" / "Parent Loop BBa_b Depth=d" / "This Loop Header"
    parent = {}  # header -> enclosing header
    member = []  # per block: innermost loop header or None
    for label, notes, body in blocks:
        txt = "\n".join(notes)
        hdr = None
        if "Loop Header" in txt and label.startswith(".LBB"):
            hdr = label[2:]
            ps = re.findall(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", txt)
            if ps:
                parent[hdr] = max(ps, key=lambda p: int(p[1]))[0]
        else:
            m = re.search(r"in Loop: Header=(BB\d+_\d+)", txt)
            if m:
                hdr = m.group(1)
        member.append(hdr)

    def inside(h, loop):
        while h is not None:
            if h == loop:
                return True
            h = parent.get(h)
        return False

    best = None
    for loop in set(h for h in member if h):
        text = "\n".join(ln for (label, notes, body), h in zip(blocks, member) if inside(h, loop) for ln in body)
        if text.count("ds_bpermute_b32") >= 8 and text.count("global_load_dwordx4") >= 6:
            if best is None or len(text) < len(best[1]):
                best = (loop, text)  # the innermost loop that holds both
    assert best, "step loop not found: the listing's shape changed (tools/step_loop_isa.py)"
    loop = best[0]
    body = []
    for (label, notes, ins), h in zip(blocks, member):
        if inside(h, loop):
            body += notes + ins
    return loop, body


def summary(body):
    ins = [ln.split(";")[0].strip() for ln in body if ln.strip() and not ln.strip().startswith((";", "."))]
    ins = [i for i in ins if i]
    kinds = {"v_": 0, "s_": 0, "ds_": 0, "global_": 0, "scratch_": 0, "buffer_": 0, "flat_": 0}
    for i in ins:
        for k in kinds:
            if i.startswith(k):
                kinds[k] += 1
    scratch = [ln.strip() for ln in body if re.match(r"\s*scratch_", ln)]
    return len(ins), kinds, scratch


if __name__ == "__main__":
    lines = function_body(device_asm())
    loop, body = step_loop(lines)
    n, kinds, scratch = summary(body)
    print(f"{FUNC}: step loop header {loop}, {n} instructions in its blocks (static count, all paths): "
          + ", ".join(f"{v} {k}*" for k, v in kinds.items() if v))
    print("scratch_ instructions inside the loop:", len(scratch))
    for s in scratch:
        print("   ", s)
    print("frame_kernel<false, false>:", kernel_resources())
    if "--dump" in sys.argv:
        with open(sys.argv[sys.argv.index("--dump") + 1], "w") as f:
            f.write(f"; gfx950 ISA of the step loop of trace_queue<1, false> (scatter rays), source_sha16 {B.source_sha16()}\n")
            f.write(f"; loop header {loop}; {n} instructions (static, all paths): " + ", ".join(f"{v} {k}*" for k, v in kinds.items() if v) + "\n")
            f.write(f"; scratch_ instructions inside the loop: {len(scratch)}\n")
            f.write("\n".join(body) + "\n")
    sys.exit(1 if scratch else 0)
