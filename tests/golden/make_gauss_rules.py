"""Generates tests/golden/gauss_rules.json from the constants of the reference's quadrature tables.

Run in the build container (the reference tree is not on the GPU box):

    python tests/golden/make_gauss_rules.py [/root/reference]

It reads `src/petigarule.c` as TEXT, finds the two tabulating functions (`IGA_Rule_GaussLegendre`, :182-319, and
`IGA_Rule_GaussLobatto`, :321-end), and evaluates the `X[i] = ...; W[i] = ...;` assignments of every `case (q):` with a
small interpreter of its own (a literal `Q(<digits>)`, a sign, or a reference to an earlier `X[j]` / `W[j]`).  What is
committed is data only: per rule and size the node / weight digit strings as the reference prints them, plus the
nearest doubles (`float()` of those strings, correctly rounded by Python).  No reference source text is kept."""
import json
import os
import re
import sys


def parse_function(text, name):
    start = text.index("static PetscErrorCode %s(PetscInt q" % name)
    start = text.index("{", start)
    end = text.index("\n}\n", start)
    body = text[start:end]
    rules = {}
    for m in re.finditer(r"case \((\d+)\):(.*?)break;", body, re.S):
        q = int(m.group(1))
        vals = {"X": {}, "W": {}}        # name -> index -> (sign, digit string)
        for a in re.finditer(r"([XW])\[(\d+)\]\s*=\s*(-?)\s*(Q\(([0-9.]+)\)|([XW])\[(\d+)\])\s*;", m.group(2)):
            arr, i, neg = a.group(1), int(a.group(2)), a.group(3) == "-"
            if a.group(5) is not None:
                sign, digits = 1, a.group(5)
            else:
                sign, digits = vals[a.group(6)][int(a.group(7))]
            vals[arr][i] = (-sign if neg else sign, digits)
        assert sorted(vals["X"]) == list(range(q)) and sorted(vals["W"]) == list(range(q)), (name, q)
        txt = lambda sd: ("-" if sd[0] < 0 and float(sd[1]) != 0.0 else "") + sd[1]
        rules[str(q)] = {
            "X_digits": [txt(vals["X"][i]) for i in range(q)],
            "W_digits": [txt(vals["W"][i]) for i in range(q)],
            "X": [float(txt(vals["X"][i])) for i in range(q)],
            "W": [float(txt(vals["W"][i])) for i in range(q)],
        }
    return rules


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    text = open(os.path.join(ref, "src", "petigarule.c")).read()
    out = {
        "provenance": "dalcinl/PetIGA src/petigarule.c: constants of IGA_Rule_GaussLegendre (q = 1..10) and "
                      "IGA_Rule_GaussLobatto (q = 2..10), extracted by tests/golden/make_gauss_rules.py",
        "legendre": parse_function(text, "IGA_Rule_GaussLegendre"),
        "lobatto": parse_function(text, "IGA_Rule_GaussLobatto"),
    }
    assert sorted(map(int, out["legendre"])) == list(range(1, 11)), sorted(out["legendre"])
    assert sorted(map(int, out["lobatto"])) == list(range(2, 11)), sorted(out["lobatto"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gauss_rules.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
