"""Provenance check, run in the build container only (it reads /root/reference).

For every python file of the package that has a same-named file in the reference, print
the difflib line ratio and every run of >= MIN_RUN consecutive identical, non-trivial
lines (imports, blank lines, decorators and bare signatures do not count).  The facade
mirrors the reference's public API, so names and signatures coincide by design; bodies
must not.

    python tools/similarity_audit.py [MIN_RUN]
"""

import difflib
import os
import sys

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scarlet_amd")
REF = "/root/reference/scarlet"
TRIVIAL_PREFIXES = ("import ", "from ", "@", "def ", "class ", '"""', "return self", "pass", ")", "]", "}")


def significant(line):
    t = line.strip()
    return bool(t) and not t.startswith("#") and not t.startswith(TRIVIAL_PREFIXES)


def audit(mine, theirs, min_run):
    a = [l.rstrip() for l in open(mine)]
    b = [l.rstrip() for l in open(theirs)]
    sa = [l.strip() for l in a]
    sb = [l.strip() for l in b]
    sm = difflib.SequenceMatcher(None, sa, sb, autojunk=False)
    runs = []
    for i, j, n in sm.get_matching_blocks():
        weight = sum(significant(x) for x in sa[i:i + n])
        if weight >= min_run:
            runs.append((i + 1, j + 1, n, weight))
    return sm.ratio(), runs


def main():
    min_run = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    worst = 0
    for root, _, files in os.walk(REPO):
        for f in sorted(files):
            if not f.endswith(".py"):
                continue
            mine = os.path.join(root, f)
            rel = os.path.relpath(mine, REPO)
            theirs = os.path.join(REF, rel)
            if not os.path.exists(theirs):
                continue
            ratio, runs = audit(mine, theirs, min_run)
            worst = max(worst, ratio)
            flag = "  <-- runs" if runs else ""
            print("%-28s ratio %.2f%s" % (rel, ratio, flag))
            for i, j, n, w in runs:
                print("      ours:%d ref:%d  %d lines (%d significant)" % (i, j, n, w))
    print("max ratio %.2f" % worst)


if __name__ == "__main__":
    main()
