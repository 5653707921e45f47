"""Audit the generated code of attention.hip: outside ;;#ASMSTART/;;#ASMEND no instruction of a backward kernel may name
an accumulator register of the range its statements own (the top `owned` AGPRs).  usage: audit_agpr.py <file.s>"""
import re, sys

OWNED = {"attn_bwd_dkv_kernelILi128ELb1E": 192, "attn_bwd_dkv_kernelILi128ELb0E": 192, "attn_bwd_dq_kernelILi128E": 128, "attn_bwd_dkv_kernelILi64ELb0E": 96, "attn_bwd_dq_kernelILi64E": 64}
OWNED_FWD2 = {"attn_fwd2_kernelILi128E": 192}  # attention_fwd2.hip


def audit(path, OWNED=OWNED):
    name, inasm, bad, seen = None, False, [], {}
    for ln, l in enumerate(open(path), 1):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            name = next((k for k in OWNED if k in m.group(1)), None)
            inasm = False
        if name is None:
            continue
        if "s_endpgm" in l:
            name = None
            continue
        if "ASMSTART" in l:
            inasm = True
        elif "ASMEND" in l:
            inasm = False
        elif inasm:
            if "v_mfma" in l:
                seen[name] = seen.get(name, 0) + 1
        else:
            lo = 256 - OWNED[name]
            for a, b in re.findall(r"\ba\[(\d+):(\d+)\]", l):
                if int(b) >= lo:
                    bad.append((name, ln, l.strip()))
            for a in re.findall(r"\ba(\d+)\b", l):
                if int(a) >= lo:
                    bad.append((name, ln, l.strip()))
    return bad, seen


if __name__ == "__main__":
    table = OWNED_FWD2 if "fwd2" in sys.argv[1] else OWNED
    bad, seen = audit(sys.argv[1], table)
    print("owned-range MFMA statements per kernel:", seen)
    for b in bad[:20]:
        print("VIOLATION", b)
    sys.exit(1 if bad or len(seen) != len(table) else 0)
