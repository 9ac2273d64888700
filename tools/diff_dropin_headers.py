#!/usr/bin/env python3
"""Build-container-only check: the public declarations of the drop-in headers
include/orbhip/ORBextractor.h / ORBmatcher.h against the reference's include/ORBextractor.h / ORBmatcher.h.

Reads /root/reference at run time, stores nothing from it, and exits 0 with "skipped" when the reference is absent
(the GPU box).  A declaration counts as
  identical   same text after removing comments and white space,
  std-only    same after also removing `std::` (the reference headers rely on `using namespace std` leaking in),
  missing     in the reference class, not in the drop-in (exit code 1),
  extra       only in the drop-in (reported; additions such as SetDevice are allowed).
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ORBHIP_REFERENCE", "/root/reference")
PAIRS = [("ORBextractor", "include/orbhip/ORBextractor.h", "include/ORBextractor.h"),
         ("ORBmatcher", "include/orbhip/ORBmatcher.h", "include/ORBmatcher.h")]


def class_body(text, name):
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    m = re.search(r"\bclass\s+%s\b[^;{]*\{" % name, text)
    if not m:
        return None
    i, depth = m.end(), 1
    while depth and i < len(text):
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def declarations(body):
    """Statements of the class body by access section, inline function bodies dropped."""
    out, access, depth, cur = {"public": [], "protected": [], "private": []}, "private", 0, ""
    i = 0
    while i < len(body):
        ch = body[i]
        if depth == 0:
            m = re.match(r"\s*(public|protected|private)\s*:", body[i:])
            if m and not cur.strip():
                access = m.group(1)
                i += m.end()
                continue
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:          # end of an inline body: the declaration is what came before it
                out[access].append(cur)
                cur = ""
                i += 1
                while i < len(body) and body[i] in " \t\r\n;":
                    i += 1
                continue
        elif depth == 0:
            if ch == ";":
                out[access].append(cur)
                cur = ""
            else:
                cur += ch
        i += 1
    norm = lambda s: re.sub(r"\s+", "", s)
    return {k: [norm(d) for d in v if norm(d)] for k, v in out.items()}


def main():
    if not os.path.isdir(REF):
        print("diff_dropin_headers: skipped (%s absent)" % REF)
        return 0
    rc = 0
    for name, mine, theirs in PAIRS:
        a = class_body(open(os.path.join(ROOT, mine)).read(), name)
        b = class_body(open(os.path.join(REF, theirs)).read(), name)
        if a is None or b is None:
            print("%s: class not found" % name)
            rc = 1
            continue
        A, B = declarations(a), declarations(b)
        nostd = lambda s: s.replace("std::", "")
        for sec in ("public", "protected"):
            mine_set = set(A[sec])
            mine_nostd = {nostd(d) for d in A[sec]}
            ident = [d for d in B[sec] if d in mine_set]
            stdonly = [d for d in B[sec] if d not in mine_set and nostd(d) in mine_nostd]
            missing = [d for d in B[sec] if d not in mine_set and nostd(d) not in mine_nostd]
            ref_nostd = {nostd(d) for d in B[sec]}
            extra = [d for d in A[sec] if nostd(d) not in ref_nostd]
            print("%-13s %-9s reference %2d: identical %2d, std-only %2d, missing %2d; extra in the drop-in %2d"
                  % (name, sec, len(B[sec]), len(ident), len(stdonly), len(missing), len(extra)))
            for d in missing:
                print("    MISSING  " + d)
                if sec == "public":
                    rc = 1
            for d in extra:
                print("    extra    " + d)
    return rc


if __name__ == "__main__":
    sys.exit(main())
