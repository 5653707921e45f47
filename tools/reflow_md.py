"""Re-wrap the paragraph lines of a Markdown file that are longer than 160 characters (tables and code blocks are left alone):  python tools/reflow_md.py DESIGN.md"""
import re
import sys

p = sys.argv[1]
L = open(p).read().split("\n")
i = 0
while i < len(L):
    l = L[i]
    if len(l) > 160 and not l.lstrip().startswith("|") and not l.startswith("```"):
        cut = l.rfind(" ", 0, 159)
        head, rest = l[:cut], l[cut + 1:]
        indent = re.match(r"\s*", l).group(0)
        if re.match(r"\s*(\*|\d+\.)\s", l):
            indent = " " * (len(indent) + (2 if l.lstrip().startswith("*") else 3))
        nxt = L[i + 1] if i + 1 < len(L) else ""
        L[i] = head
        if nxt.strip() and not nxt.lstrip().startswith(("|", "*", "#")) and not re.match(r"\s*\d+\.\s", nxt) and len(re.match(r"\s*", nxt).group(0)) == len(indent):
            L[i + 1] = indent + rest + " " + nxt.lstrip()
        else:
            L.insert(i + 1, indent + rest)
    i += 1
open(p, "w").write("\n".join(L))
print(max(len(x) for x in L if not x.lstrip().startswith("|")))
