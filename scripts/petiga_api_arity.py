#!/usr/bin/env python3
"""Generates tests/golden/petiga_api_arity.json from the reference's public header (include/petiga.h): the argument count of
every PETSC_EXTERN function and function-like macro, and the field names of struct _p_IGA / _n_IGAAxis / _n_IGABasis / _n_IGARule /
_n_IGAForm.  Data about the interface, not its text: tests/test_adapter_signatures.py checks every PetIGA call and every struct
field the adapter (adapter/petiga_amd_petsc.c, never compiled here: no PETSc) uses against it."""
import json
import os
import re
import sys


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def parse_header(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    funcs, macros, structs = {}, {}, {}
    for m in re.finditer(r"PETSC_EXTERN\s+[\w\s\*]+?\b(\w+)\s*\(([^;{]*?)\)\s*;", text):
        args = split_args(m.group(2))
        funcs[m.group(1)] = 0 if args == ["void"] else len(args)
    for m in re.finditer(r"^[ \t]*#\s*define\s+(\w+)\(([^)]*)\)", text, flags=re.M):
        macros[m.group(1)] = len(split_args(m.group(2)))
    for m in re.finditer(r"struct\s+(_[pn]_\w+)\s*\{(.*?)\n\};", text, flags=re.S):
        body = re.sub(r"\([^)]*\)\s*\([^)]*\)", " ", m.group(2))      # function-pointer members
        names = set()
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl or decl.startswith("#"):
                continue
            decl = re.sub(r"\[[^\]]*\]", "", decl)
            for part in decl.split(","):
                w = re.findall(r"[A-Za-z_]\w*", part)
                if w:
                    names.add(w[-1])
        structs[m.group(1)] = sorted(names)
    return dict(functions=funcs, macros=macros, structs=structs)


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/include/petiga.h"
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "petiga_api_arity.json")
    data = parse_header(open(ref).read())
    data["source"] = "include/petiga.h of dalcinl/PetIGA (argument counts and field names only)"
    json.dump(data, open(out, "w"), indent=0, sort_keys=True)
    print(out, len(data["functions"]), "functions", len(data["macros"]), "macros", {k: len(v) for k, v in data["structs"].items()})
