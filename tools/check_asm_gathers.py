"""Build-time check of the split-pair plane-sweep kernel's inline-asm gathers (conv0_sweep_x3.hip, cooperative producers).

The sweep kernels keep `global_load_dwordx4` instructions issued from inline asm in flight across loop iterations and count them by
hand.  For the compiler an asm output is final the moment the asm statement is issued, so nothing in the language stops it from
(a) splitting the live range of such a register (a v_mov copy made while the load has not landed: the copy is stale, and the freed
register is reused under the landing load) or (b) sinking a consumer below the re-request and keeping "the old value" in a copy
made before the wait.  Both happened while the cooperative producers of conv0_sweep_x3.hip were written (a memory fault from wild
gather offsets; one voxel group wrong from run to run).  The source is now shaped so that hipcc has no reason to do either; this
script checks the ISA it actually produced:
  * every gather destination tuple is used by the same number of gather instructions (prologue + loop body name the same registers),
  * no v_mov reads or writes a gather destination inside the producer region.
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
    findings = []
    for name in re.findall(r'^(_Z\S*conv0_sweep_x3\S*):', asm_text, re.M):      # the 16-bit kernel shares registers between its roles: not checkable this way
        a = asm_text.index(name + ':')
        body = asm_text[a:asm_text.index('.Lfunc_end', a)].split('\n')
        gl = [(i, l.split()[1].rstrip(',')) for i, l in enumerate(body) if 'global_load_dwordx4' in l and ', s[' in l]
        cnt = collections.Counter(t for _, t in gl)
        rolling = {t: c for t, c in cnt.items() if c >= 2}
        if not rolling:
            continue
        if len(set(rolling.values())) != 1:
            findings.append(f"{name[:50]}: gather destinations are not used uniformly: {sorted(rolling.items())}")
        dest = set()
        for t in rolling:
            dest |= regs(t)
        lo = min(i for i, t in gl if t in rolling)
        hi = max(i for i, t in gl if t in rolling)
        for i in range(max(lo - 5, 0), min(hi + 60, len(body))):
            l = body[i].strip()
            if l.startswith('v_mov') and any(regs(o) & dest for o in l.split()[1:]):
                findings.append(f"{name[:50]}: line {i}: {l}")
        print(f"{name[:60]}: {len(rolling)} rolling gather destinations x {set(rolling.values())} uses, {len(dest)} registers checked")
    return findings


if __name__ == "__main__":
    src = sys.argv[1]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-S",
                               "--cuda-device-only", os.path.abspath(src), "-o", out] + sys.argv[2:], cwd=os.path.dirname(os.path.abspath(src)))
        f = check(open(out).read())
    for x in f:
        print("FINDING:", x)
    sys.exit(1 if f else 0)
