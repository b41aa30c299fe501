"""Build-time check of the plane-sweep kernels' inline-asm gathers (conv0_sweep.hip, conv0_sweep_x3.hip).

The sweep kernels keep `global_load_dwordx4` instructions issued from inline asm in flight across loop iterations and count them by
hand.  For the compiler an asm output is final the moment the asm statement is issued, so nothing in the language stops it from
(a) splitting the live range of such a register (a v_mov copy made while the load has not landed: the copy is stale, and the freed
register is reused under the landing load) or (b) sinking a consumer below the re-request and keeping "the old value" in a copy
made before the wait.  Both happened while the cooperative producers of conv0_sweep_x3.hip were written (a memory fault from wild
gather offsets; one voxel group wrong from run to run).  The source is now shaped so that hipcc has no reason to do either; this
script checks the ISA it actually produced:
  * the prologue gathers name destination tuples that the plane loop names too (no re-homing between prologue and loop),
  * no v_mov reads or writes a gather destination inside the producers' plane loop.
usage: check_asm_gathers.py <file.hip> [extra hipcc flags...]   (exit code 1 on a finding)"""
import collections
import os
import re
import subprocess
import sys
import tempfile


def regs(tok):
    tok = tok.rstrip(',')
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def check(asm_text):
    """For every conv0_sweep* kernel: the loop that holds the rolling gathers (>= 16 saddr-form global_load_dwordx4) must not contain
    a v_mov that reads or writes one of their destination registers, and every prologue gather (same form, outside the loop) must
    name a destination tuple that the loop names too."""
    findings = []
    for name in re.findall(r'^(_Z\S*conv0_sweep\S*):', asm_text, re.M):
        a = asm_text.index(name + ':')
        body = asm_text[a:asm_text.index('.Lfunc_end', a)].split('\n')
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        is_gather = lambda l: 'global_load_dwordx4' in l and ', s[' in l
        cands = [(y - x, x, y) for x, y in loops if sum(is_gather(l) for l in body[x:y + 1]) >= 16]
        if not cands:
            continue
        # the innermost such loop is the producers' plane loop.  "Innermost" by LLVM's own loop-depth comment on the header label (hipcc may lay
        # the body of an inner loop out BEHIND the back edge of its parent, so that the two line ranges are disjoint and the shorter one is the
        # parent's prologue); ties: the shorter range
        def depth(x):
            for l in body[x:x + 3]:              # "=>This Inner Loop Header: Depth=2" or "in Loop: Header=BB0_37 Depth=1", on or right under the label
                m = re.search(r'Depth=(\d+)', l)
                if m:
                    return int(m.group(1))
            return 0
        _, _, x, y = min((-depth(x), span, x, y) for span, x, y in cands)
        inside = [l.split()[1].rstrip(',') for l in body[x:y + 1] if is_gather(l)]
        dest = set()
        for t in inside:
            dest |= regs(t)
        for i in range(x, y + 1):
            l = body[i].strip()
            if l.startswith('v_mov') and any(regs(o) & dest for o in l.split()[1:]):
                findings.append(f"{name[:50]}: line {i}: {l}")
        # (round 3) no data-dependent control flow inside the plane loop: a wave-uniform branch around the blend (skip the arithmetic of
        # a wave whose voxels all project outside the partner image) brought the run-to-run corruption back in a build-dependent
        # way — one build re-homed gather registers at the join (caught above), another passed every register check and still
        # produced wrong voxels until every counted wait was replaced by vmcnt(0).  Only the loop's back edge and EXEC-mask skips
        # (s_cbranch_execz / execnz around `if (active lane)`) may branch here.
        for i in range(x, y):
            l = body[i].strip()
            m = re.match(r's_cbranch_(?:vcc|scc)\w*\s+(\.LBB\d+_\d+)', l)
            if m and x < labels.get(m.group(1), -1) <= y:          # target inside the loop and not its header: not the loop control
                findings.append(f"{name[:50]}: line {i}: data-dependent branch inside the producers' plane loop: {l}")
        # prologue gathers: the 16 / 32 requests in front of the loop (the last saddr gathers before it)
        before = [l.split()[1].rstrip(',') for l in body[max(0, x - 400):x] if is_gather(l)]
        pro = [t for t in before if regs(t) & dest]
        stray = [t for t in pro if t not in set(inside)]
        if stray:
            findings.append(f"{name[:50]}: prologue gathers into tuples the loop does not use: {stray}")
        # (round 5) the persistent sweep kernel keeps the gathers rolling ACROSS tiles: the last plane of a tile is peeled out of the plane loop
        # (its requests are the next tile's), so the rolling tuples are also written in the tile loop around it.  Anywhere behind the first
        # request into a rolling tuple, up to the end of the kernel, no move / AGPR copy / scratch access may name one of its registers.
        first = {}
        in_asm = False
        for i, l in enumerate(body):
            if 'ASMSTART' in l:
                in_asm = True
            elif 'ASMEND' in l:
                in_asm = False
            if in_asm and is_gather(l):      # inline-asm requests only (the compiler's own loads reuse the register names)
                t = l.split()[1].rstrip(',')
                if t in set(inside):
                    first.setdefault(t, i)
        outside = 0
        for i, l in enumerate(body):
            ls = l.strip()
            if not (x <= i <= y) and ls.startswith(('v_mov', 'v_accvgpr', 'scratch_', 'v_swap')):
                touched = set()
                for o in ls.split()[1:]:
                    touched |= regs(o)
                for t, fi in first.items():
                    if i > fi and touched & regs(t):
                        findings.append(f"{name[:50]}: line {i} (outside the plane loop, behind the first request into {t}): {ls}")
                        outside += 1
        print(f"{name[:60]}: plane loop lines {x}..{y}, {len(inside)} gathers in the loop into {len(set(inside))} tuples, "
              f"{len(pro)} prologue gathers, {len(dest)} registers checked, {len(first)} tuples followed to the end of the kernel")
    return findings


if __name__ == "__main__":
    src = sys.argv[1]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        # the flags of rgbmanip_amd/csrc/build.sh (incl. no packed-fp32 instructions): the ISA that is checked is the ISA that ships
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Xclang", "-target-feature",
                               "-Xclang", "-packed-fp32-ops", "-S",
                               "--cuda-device-only", os.path.abspath(src), "-o", out] + sys.argv[2:], cwd=os.path.dirname(os.path.abspath(src)))
        f = check(open(out).read())
    for x in f:
        print("FINDING:", x)
    sys.exit(1 if f else 0)
