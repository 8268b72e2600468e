"""Static screen of built gfx950 ISA for the hazards an inline-asm MFMA hides from the compiler.

The compiler's hazard recognizer does not look inside `asm volatile("v_mfma ...")`: it neither knows that the statement reads its accumulator
nor that it reads its A / B operands as an MFMA does, so the wait states it would insert for a builtin MFMA are missing:
  (1) a VALU instruction (v_*; includes v_accvgpr_read / write and v_mov copies the register allocator makes) that writes a register an MFMA
      reads must be at least TWO wait states in front of it;
  (2) an MFMA that reads as SrcC an accumulator written by an MFMA fewer than two instructions earlier (dependent back-to-back issue);
Both produced silently wrong numbers in this code base (profiles/round3_lstm_ordering.txt).  Usage:
    python tools/check_mfma_hazards.py file.s [kernel-name-substring ...]   -> prints findings, exit status 1 if any."""
import re
import sys

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def kernels(text):
    cur, name = None, None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                yield name, cur
                cur = None
            elif line.startswith("\t") and not line.strip().startswith((".", ";")):
                cur.append(line.strip().split(";")[0].strip())
            elif re.match(r"^\.LBB", line):
                cur.append("LABEL")


def check(name, ins):
    bad = []
    for i, l in enumerate(ins):
        if not l.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
        dst, srcs = regs(ops[0]), set().union(*[regs(o) for o in ops[1:4]])
        srcc = regs(ops[3]) if len(ops) > 3 else set()
        waited, k, mf = 0, i - 1, 0
        while k >= 0 and waited < 2:
            p = ins[k]
            if p == "LABEL":
                break                                   # (a branch target: whatever precedes is another path; labels inside the product loops are checked by eye)
            if p.startswith("s_nop"):
                waited += int(p.split()[1]) + 1
            else:
                if p.startswith("v_") and not p.startswith("v_mfma"):
                    w = regs(p.split(None, 1)[1].split(",")[0]) if " " in p else set()
                    if w & srcs:
                        bad.append("%s: VALU write %d wait state(s) in front of an MFMA that reads it:\n      %s\n      %s" % (name, waited, p, l))
                if p.startswith("v_mfma"):
                    pd = regs(p.split(None, 1)[1].split(",")[0])
                    if pd & srcc and mf < 1:
                        bad.append("%s: dependent MFMAs %d apart:\n      %s\n      %s" % (name, mf + 1, p, l))
                    mf += 1
                waited += 1
            k -= 1
    return bad


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2:]
    found = []
    n = 0
    for name, ins in kernels(text):
        if want and not any(w in name for w in want):
            continue
        n += 1
        found += check(name, ins)
    print("%d kernel(s) scanned, %d finding(s)" % (n, len(found)))
    for f in found[:40]:
        print("  " + f)
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
