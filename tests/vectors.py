"""Loaders for the committed golden fixtures (tests/golden/*.json)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def b(s: str) -> bytes:
    return s.encode("latin1")


def tup(pairs):
    return [tuple(p) for p in pairs]


def testcc():
    return _load("testcc_vectors.json")


def semantics():
    return _load("semantics_vectors.json")


def fuzz():
    return _load("fuzz_vectors.json")


def bench():
    return _load("bench_vectors.json")


def artefact():
    return _load("artefact_vectors.json")


def artefact_cases():
    """(regex, text, the reference's MatchAll offsets, MatchFull) where the reference's ring artefact applies or
    nearly applies (make_golden.py: gen_artefact) -- expectations from the real reference, never from the oracle."""
    for v in artefact():
        yield b(v["regex"]), b(v["text"]), tup(v["ref_all"]), v["ref_full"]


def all_matchall_cases():
    """Every (regex, text, expected MatchAll offsets, expected MatchFull) in the fixtures."""
    seen = set()
    for v in testcc():
        key = (v["regex"], v["text"])
        if key in seen:
            continue
        seen.add(key)
        yield b(v["regex"]), b(v["text"]), tup(v["ref_all"]), v["ref_full"]
    for v in semantics()["vectors"]:
        yield b(v["regex"]), b(v["text"]), tup(v["ref_all"]), v["ref_full"]
    for v in fuzz():
        yield b(v["regex"]), b(v["text"]), tup(v["ref_all"]), v["ref_full"]
