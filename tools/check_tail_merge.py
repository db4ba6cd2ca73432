"""Static check of the merge-tail hand-off (csrc/tail_merge.h) on the gfx950 assembly of every kernel that carries it
(K1: scan_f32.hip / scan_f16.hip; the per-query IVF scans: ivf.hip).

The protocol has no release/acquire fence pair (measured +18 us per launch); it relies on three properties of the
GENERATED code, which a compiler upgrade could silently break, so the build checks them:

  1. every 8-byte store in front of the ticket atomic - the block's published k-list - is a device-scope,
     write-through store (`global_store_dwordx2 ... sc1`), and there is at least one;
  2. an `s_waitcnt vmcnt(0)` stands between the last such store and the ticket (`global_atomic_add ... sc0`, the
     returning add of tail_last_block): the list has left the CU before the ticket can be seen;
  3. every 8-byte load behind the ticket - the last block reading the other blocks' lists - bypasses this CU's L1
     (`global_load_dwordx2 ... sc1`); a plain or flat 8-byte load there could return a stale line.

The listing of a kernel is walked in text order: publish, ticket and merge are straight-line in the source and the
compiler lays their blocks out in that order (the check fails loudly if it ever does not).

usage: check_tail_merge.py file.s     (exit status 1 if anything is reported)"""
import re
import sys


def kernels(path):
    kern, cur = {}, None
    for i, l in enumerate(open(path).read().split("\n")):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1); kern[cur] = []
        elif l.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            kern[cur].append((i + 1, l.strip()))
    return kern


def main(path):
    total = checked = 0
    for name, lines in kernels(path).items():
        ev = []
        for n, t in lines:
            if not t or t[0] in ";.#":
                continue
            op = t.split()[0]
            sc1 = bool(re.search(r"\bsc1\b", t))
            if op == "global_atomic_add" and re.search(r"\bsc0\b", t):
                ev.append(("ticket", n, t, True))
            elif op in ("global_store_dwordx2", "flat_store_dwordx2", "buffer_store_dwordx2"):
                ev.append(("store8", n, t, sc1 and op == "global_store_dwordx2"))
            elif op in ("global_load_dwordx2", "flat_load_dwordx2", "buffer_load_dwordx2"):
                ev.append(("load8", n, t, sc1 and op == "global_load_dwordx2"))
            elif op == "s_waitcnt" and re.search(r"vmcnt\(0\)", t):
                ev.append(("wait0", n, t, True))
        tickets = [i for i, e in enumerate(ev) if e[0] == "ticket"]
        if not tickets:
            continue
        checked += 1
        last = tickets[-1]
        bad = []
        stores = [i for i, e in enumerate(ev[:last]) if e[0] == "store8"]
        if not stores:
            bad.append((ev[last][1], "no published list (8-byte store) in front of the ticket"))
        bad += [(ev[i][1], "published list stored without sc1: " + ev[i][2]) for i in stores if not ev[i][3]]
        if stores and not any(e[0] == "wait0" for e in ev[stores[-1] + 1:last]):
            bad.append((ev[last][1], "no s_waitcnt vmcnt(0) between the last published store and the ticket"))
        bad += [(e[1], "8-byte load behind the ticket without sc1: " + e[2]) for e in ev[last + 1:] if e[0] == "load8" and not e[3]]
        short = re.sub(r"^_ZN4ssmi\d+", "", name)[:72]
        print(f"{short}: {len(stores)} published stores, {sum(1 for e in ev[last + 1:] if e[0] == 'load8')} merge loads, problems: {len(bad)}")
        for n, t in bad[:8]:
            print("    line", n, t)
        total += len(bad)
    print(f"{checked} kernels with a merge tail checked")
    return 1 if (total or not checked) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
