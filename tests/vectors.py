"""Loaders for the committed golden fixtures (tests/golden/*.json)."""
import base64
import hashlib
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


def highbyte():
    return _load("highbyte_vectors.json")


class Digest:
    """An expected match list kept as count + sha256 of repr([(begin, end), ...]) (make_golden.py): equal to a list of
    pairs with that count and digest."""

    def __init__(self, d):
        self.count, self.sha256 = d["count"], d["sha256"]

    def __eq__(self, other):
        if isinstance(other, Digest):
            return (self.count, self.sha256) == (other.count, other.sha256)
        return (isinstance(other, list) and len(other) == self.count and
                hashlib.sha256(repr([tuple(p) for p in other]).encode()).hexdigest() == self.sha256)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __repr__(self):
        return "Digest(count=%d, sha256=%s...)" % (self.count, self.sha256[:12])


def highbyte_cases(max_text=None):
    """(regex, text, the reference's MatchAll offsets -- a list, or a Digest of a long one --, MatchFull) with bytes >= 0x80
    in pattern and / or text (make_golden.py: gen_highbyte); max_text: only the texts up to that many bytes."""
    for v in highbyte():
        tx = base64.b64decode(v["text_b64"])
        if max_text is not None and len(tx) > max_text:
            continue
        yield b(v["regex"]), tx, (tup(v["ref_all"]) if "ref_all" in v else Digest(v["ref_digest"])), v["ref_full"]


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
    # bytes >= 0x80 in pattern and / or text: the short texts here (every consumer, incl. the lane-by-lane CPU emulations);
    # all of them in highbyte_cases() (oracle, lowering, GPU)
    for case in highbyte_cases(max_text=40):
        yield case
