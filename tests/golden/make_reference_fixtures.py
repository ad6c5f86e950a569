#!/usr/bin/env python3
"""Extract the DATA (byte arrays, witness maps, oracle call payloads) held by the reference's own tests
into tests/golden/reference_vectors.json. Run in the build container only (reads /root/reference):

    python tests/golden/make_reference_fixtures.py

Sources:
  /root/reference/acir/tests/test_program_serialization.rs   7 byte-exact gzip+bincode circuits
  /root/reference/acvm_js/test/shared/*.ts                   bytecode + initial witness + expected witness
Nothing but literal data is copied; no reference source text is stored.
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")


def ints(txt):
    return [int(x) for x in re.findall(r"\d+", txt)]


def rust_serializations():
    src = open(f"{REF}/acir/tests/test_program_serialization.rs").read()
    out = {}
    for m in re.finditer(r"fn (\w+)\(\)\s*\{(.*?)\n\}", src, re.S):
        name, body = m.group(1), m.group(2)
        v = re.search(r"expected_serialization: Vec<u8> = vec!\[(.*?)\];", body, re.S)
        if v:
            out[name] = ints(v.group(1))
    return out


def parse_map(txt):
    return {int(k): v for k, v in re.findall(r"\[\s*(\d+),\s*\"(0x[0-9a-f]+)\"\s*\]", txt)}


def parse_nested_hex(txt):
    """oracleCallInputs / oracleResponse: arrays of hex strings or arrays of arrays."""
    txt = re.sub(r"\s+", "", txt)
    txt = txt.replace(",]", "]")
    return json.loads(txt)


def ts_fixtures():
    out = {}
    d = f"{REF}/acvm_js/test/shared"
    for fn in sorted(os.listdir(d)):
        src = open(os.path.join(d, fn)).read()
        name = fn[:-3]
        fx = {}
        for m in re.finditer(r"export const (\w+)(?::\s*\w+)?\s*=\s*(.*?);\n", src, re.S):
            key, val = m.group(1), m.group(2)
            if val.startswith("Uint8Array.from"):
                fx[key] = ints(val[len("Uint8Array.from"):])
            elif val.startswith("new Map(["):
                fx[key] = parse_map(val)
            elif val.startswith("new Map(initialWitnessMap).set("):
                base = dict(fx["initialWitnessMap"])
                k, v = re.search(r"\.set\(\s*(\d+),\s*\"(0x[0-9a-f]+)\"", val, re.S).groups()
                base[int(k)] = v
                fx[key] = base
            elif val.lstrip().startswith("["):
                fx[key] = parse_nested_hex(val)
            elif val.lstrip().startswith('"'):
                fx[key] = json.loads(val.strip())
            else:
                fx[key] = int(val.strip())
        out[name] = fx
    return out


if __name__ == "__main__":
    data = {"serialization": rust_serializations(), "acvm_js": ts_fixtures()}
    with open(OUT, "w") as f:
        json.dump(data, f, indent=0, sort_keys=True)
    print("wrote", OUT, {k: list(v) for k, v in data.items()})
