"""Static check of the register-ring kernels (mfma_r128.hip, ivf_lists_mfma.hip) on their gfx950 assembly.

The VGPRs that the opaque `global_load_dwordx4` statements name are written by the memory system long after
the instruction; until the counted wait of that ring slot no instruction may read or write them (a register
copy inserted by the compiler would copy stale bits, a reuse would be overwritten when the data lands).  The
compiler knows nothing of this, so the build checks its output: the listing is walked in text order (the
pipelined loop is fully unrolled, so text order is issue order), every opaque load marks its destination
registers as pending, every counted wait `s_waitcnt vmcnt(N>0)` releases the oldest four loads (one ring slot),
`s_waitcnt vmcnt(0)` releases all; any other instruction that touches a pending register is reported.

usage: check_ring_regs.py file.s     (exit status 1 if anything is reported)"""
import re, sys

def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def main(path):
    kern, cur = {}, None
    for i, l in enumerate(open(path).read().split("\n")):
        m = re.match(r"^(_ZN4ssmi\S+):", l)
        if m: cur = m.group(1); kern[cur] = []
        elif l.startswith(".Lfunc_end"): cur = None
        elif cur: kern[cur].append((i + 1, l))
    total = 0
    for name, lines in kern.items():
        pending, bad, inasm, nload = [], [], False, 0
        for n, l in lines:
            if "#ASMSTART" in l: inasm = True; continue
            if "#ASMEND" in l: inasm = False; continue
            t = l.strip()
            if not t or t[0] in ";.": continue
            parts = t.replace(",", " ").split()
            op, ops = parts[0], parts[1:]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m and int(m.group(1)) == 0: pending = []
                elif m and inasm: pending = pending[4:]
                continue
            if inasm and op == "global_load_dwordx4":
                pending.append(regs(ops[0])); nload += 1
                # the address operand may share registers with the destination (read at issue): fine
                continue
            live = set().union(*pending) if pending else set()
            if any(regs(o) & live for o in ops): bad.append((n, t))
        short = re.sub(r"^_ZN4ssmi\d+", "", name)[:64]
        print(f"{short}: {nload} opaque loads, instructions touching an in-flight ring register: {len(bad)}")
        for n, t in bad[:8]: print("    line", n, t)
        total += len(bad)
    return 1 if total else 0

if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
