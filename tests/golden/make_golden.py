#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ -- run ONLY in the build
container, where /root/reference exists and oracle/_ref has been built:

    make -C oracle ref && python tests/golden/make_golden.py

What it produces (all of it is DATA: inputs and expected outputs; no reference source
text is stored):

  testcc_vectors.json    every TEST* line of the reference's tools/tests/test.cc:193-534
                         (regexp, text, the expectation written in test.cc itself) expanded
                         the way its macros expand them (TEST_Full / TEST_Multiple /
                         33-alignment TEST_Multiple_unbound, test.cc:175-190,665-715), plus
                         the outputs of the real reference run here with
                         use_fast_forward=0 (MatchAll offsets, MatchFull) -- the oracle
                         configuration of SURVEY.md section 8c.
  semantics_vectors.json hand-picked probes (the x? == x* quirk,
                         Q1-Q3 inputs) with reference(ff=0) outputs and, where they differ,
                         the known-bad default-flag outputs.
  fuzz_vectors.json      seeded random (regexp, text) pairs with reference(ff=0) outputs.
  artefact_vectors.json  inputs on which the reference's ring artefact ("Q8") applies or nearly
                         applies: hand-written at-risk patterns and random patterns on which the reference and
                         the documented semantics were seen to differ, over texts of up to 600 bytes with many
                         adjacent candidates; reference(ff=0) outputs.  (The oracle's `match_all_spec` is used
                         only to SELECT inputs; every expectation stored is the real reference's.)
  highbyte_vectors.json  bytes >= 0x80 in patterns and texts (round 5): signed bracket ranges around 0x7f / 0x80 (the
                         reference compares bytes as signed chars), `.` / \\S / \\D / negated classes over Latin-1 and
                         UTF-8 text, literal windows of high bytes, random patterns over alphabets with high bytes;
                         texts of up to 5000 bytes; reference(ff=0) outputs.
  bench_vectors.json     the 12 benchmark regexps (tools/benchmarks/run.py:347-360) on
                         seeded random text with planted matches; regexdna patterns on the
                         FASTA n=50000 input; reference(ff=0) outputs (offsets or digest).

The reference is driven in a child process with a timeout: it can abort or loop forever
on inputs its own tests never exercise.
"""
import base64
import hashlib
import json
import multiprocessing as mp
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from checkers import Ref  # noqa: E402
from rejit_amd import workloads as W  # noqa: E402

REF_TEST_CC = "/root/reference/tools/tests/test.cc"


# --------------------------------------------------------------------------- ref worker
def _worker(conn):
    ref0 = Ref(use_ff=0)
    while True:
        msg = conn.recv()
        if msg is None:
            return
        kind, flags, regex, text = msg
        ref0.set_flags(*flags)
        if kind == "all":
            conn.send(ref0.match_all(regex, text))
        elif kind == "full":
            conn.send(ref0.match_full(regex, text))
        elif kind == "first":
            conn.send(ref0.match_first(regex, text))


class RefProc:
    """The reference in a child process; returns 'crash' / 'timeout' instead of dying."""
    FF0 = (0, 0, 1, 1)
    DEFAULT = (1, 1, 1, 1)

    def __init__(self):
        self.p = None

    def _start(self):
        self.parent, child = mp.Pipe()
        self.p = mp.Process(target=_worker, args=(child,), daemon=True)
        self.p.start()

    def call(self, kind, regex, text, flags=FF0, timeout=10.0):
        if self.p is None or not self.p.is_alive():
            self._start()
        self.parent.send((kind, flags, regex, text))
        if self.parent.poll(timeout):
            try:
                return self.parent.recv()
            except EOFError:
                self.p = None
                return "crash"
        alive = self.p.is_alive()
        self.p.kill()
        self.p = None
        return "timeout" if alive else "crash"


# --------------------------------------------------------------------------- test.cc
def _c_string_expr(src: str) -> bytes:
    """Evaluate `"lit" "lit" x10("lit") ...` (test.cc:102-104 defines x10/x50/x100)."""
    pos = 0
    out = b""

    def skip_ws():
        nonlocal pos
        while pos < len(src) and src[pos].isspace():
            pos += 1

    def term():
        nonlocal pos
        skip_ws()
        if src[pos] == '"':
            pos += 1
            buf = []
            while src[pos] != '"':
                if src[pos] == "\\":
                    buf.append(src[pos:pos + 2])
                    pos += 2
                else:
                    buf.append(src[pos])
                    pos += 1
            pos += 1
            return "".join(buf).encode("latin1").decode("unicode_escape").encode("latin1")
        m = re.match(r"x(10|50|100)\(", src[pos:])
        assert m, src[pos:]
        pos += m.end()
        inner = b""
        while True:
            skip_ws()
            if src[pos] == ")":
                pos += 1
                break
            inner += term()
        return inner * int(m.group(1))

    while True:
        skip_ws()
        if pos >= len(src):
            return out
        out += term()


def _split_args(s: str):
    args, depth, cur, in_str, i = [], 0, "", False, 0
    while i < len(s):
        ch = s[i]
        if in_str:
            cur += ch
            if ch == "\\":
                cur += s[i + 1]
                i += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
            cur += ch
        elif ch == "(":
            depth += 1
            cur += ch
        elif ch == ")":
            depth -= 1
            cur += ch
        elif ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    args.append(cur.strip())
    return args


def parse_test_cc():
    tests = []
    for lineno, line in enumerate(open(REF_TEST_CC, encoding="latin1"), 1):
        m = re.match(r"\s*(TEST|TEST_Full|TEST_Multiple|TEST_Multiple_unbound)\((.*)\);\s*(//.*)?$", line)
        if not m or lineno < 190:
            continue
        kind, args = m.group(1), _split_args(m.group(2))
        if kind == "TEST":
            mt, exp, rx, tx = args
            tests.append(dict(line=lineno, macro=kind, match_type=mt, expected=int(exp),
                              regex=_c_string_expr(rx), text=_c_string_expr(tx)))
        elif kind == "TEST_Full":
            exp, rx, tx = args
            tests.append(dict(line=lineno, macro=kind, expected=int(exp),
                              regex=_c_string_expr(rx), text=_c_string_expr(tx)))
        else:
            exp, rx, tx, st, en = args
            tests.append(dict(line=lineno, macro=kind, expected=int(exp), start=int(st), end=int(en),
                              regex=_c_string_expr(rx), text=_c_string_expr(tx)))
    return tests


def s(b: bytes) -> str:
    return b.decode("latin1")


def pairs(x):
    return [list(p) for p in x] if isinstance(x, list) else x


def gen_testcc(ref: RefProc):
    out = []
    for t in parse_test_cc():
        rx = t["regex"]
        texts = []
        if t["macro"] == "TEST_Multiple_unbound":
            for i in range(33):  # test.cc:687-712: i leading and 32-i trailing spaces
                texts.append((i, b" " * i + t["text"] + b" " * (32 - i)))
        else:
            texts.append((0, t["text"]))
        for shift, text in texts:
            allm = ref.call("all", rx, text)
            full = ref.call("full", rx, text)
            assert isinstance(allm, list) and full in (0, 1), (t, allm, full)
            rec = dict(line=t["line"], macro=t["macro"], regex=s(rx), text=s(text), shift=shift,
                       ref_all=pairs(allm), ref_full=full)
            # expectations written in test.cc itself
            if t["macro"] == "TEST":
                rec["match_type"] = t["match_type"]
                rec["expected"] = t["expected"]
            elif t["macro"] == "TEST_Full":
                rec["expected_full"] = t["expected"]
            else:
                rec["expected_count"] = t["expected"]
                if t["expected"]:
                    rec["expected_first"] = [t["start"] + shift, t["end"] + shift]
            dflt = ref.call("all", rx, text, flags=RefProc.DEFAULT)
            if dflt != allm:
                rec["known_bad_reference_default"] = pairs(dflt)
            out.append(rec)
    return out


# --------------------------------------------------------------------------- semantics
SEMANTICS = [
    ("ab+", "xabbbx_ababx"), ("ab?", "xabx_ax"), ("ab*", "xabbbx"), ("ab{2}", "xabbx_ababx"),
    ("\\x4a", "J@"), ("\\x4A", "J@"), ("(a|ab)", "xab_"), ("(a|ab)(c|bcd)", "abcd"),
    ("x*", "aaxxa"), ("[a-c-e]", "a-e_d"), ("^", "a\r\nb\n"), ("$", "a\r\nb\n"),
    # the reference treats every repetition with max == 1 as unbounded (codegen.cc:266-312)
    ("a?", "aaa"), ("(ab)?", "ababab"), ("xa?y", "xaay_xay_xy"), ("a{1}", "aaa"), ("a{1,1}", "aaa"),
    ("xa{0,1}y", "xaay_xay_xy"), ("(ab){2,3}", "abababab"), ("x(ab){2,3}y", "xababababy"),
    ("[ab]{2,3}", "ababab"), ("[ab]{0,1}c", "abc"), ("x[ab]?y", "xaay_xay"),
    ("a{0}", "xay"), ("a{0,0}b", "xaby"), ("a**", "xaay"), ("(a*)*", "xaay"), ("(a|b*)*", "xaaby"),
    ("$*", "ab"), ("(^a|b$)+", "ab\nab"), ("^*a", "xay"), ("a{,2}", "xaay)aaaaaa.J@"),
    ("a{2}{3}", "xaay)aaaaaa.J@"), (")", "xaay)a"), ("a)", "xa)"), ("[a-]b", "xab_-b_a-b"),
    ("[-a]b", "x-b_ab"), ("[^-a]b", "x-b_ab_cb"), ("[.]", "a.b"), ("[\\d]", "a\\d1"), ("[^]", "ab"),
    ("\\d+", "ab123cd4"), ("\\D+", "ab123cd4"), ("\\s+x", "a \t x"), ("\\S+", "ab cd\tef"),
    ("a\\n*", "xa\n\na\n"), ("a\\t+b", "a\tb"), ("\\(\\)\\{\\}\\[\\]\\|\\*\\+\\^\\$\\\\", "(){}[]|*+^$\\"),
    ("a{2,}", "a_aa_aaa_aaaa"), ("(ab|a)(bc|c)?", "abc_ac_abcc"), (".*", "ab\ncd\r\n"),
    (">.*\n|\n", ">h1\nacgt\nac\n>h2\ngg\n"), ("a|b|c", "xcbay"), ("(a|b)*c", "ababc_c_abx"),
    ("ab|abc|abcd", "abcdabcab"), ("(abcd|_....efgh)", "_abcdefgh"),
    # Q1 / Q2 / Q3 inputs: default flags are wrong on these
    ("agggtaaa|tttaccct", "agggtaaa"),
    ("([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)",
     "xxcomplexregexpabcdefghdas well__regexpregexpabcdefghthe___"),
    ("regexp", "x" * 70 + "\x00" + "xxxx" + "regexp" + "y" * 50 + "regexp" + "z" * 60),
    ("ab(c.e.g)?|dexy", "abcdexyz"),
]

PARSE_ERRORS = ["a{2", "a{2,1}", "a\\", "a\\.", "a\\?", "a{}", "a{x}", "a{1,x}", "a{1,2"]
# patterns on which the reference aborts / asserts / reads out of bounds; the product
# rejects them with ParserError (documented divergence)
REJECTED = ["", "(", "(a", "|a", "a|", "a||b", "()", "(|a)", "*", "+a", "?", "{2}", "a|*", "(*a)",
            "a]b", "[a", "[", "[^", "[a-", "\\x4", "\\xZZ", "\\x"]


def gen_semantics(ref: RefProc):
    out = []
    for rx, tx in SEMANTICS:
        rxb, txb = rx.encode("latin1"), tx.encode("latin1")
        allm = ref.call("all", rxb, txb)
        full = ref.call("full", rxb, txb)
        assert isinstance(allm, list), (rx, allm)
        rec = dict(regex=rx, text=tx, ref_all=pairs(allm), ref_full=full)
        dflt = ref.call("all", rxb, txb, flags=RefProc.DEFAULT, timeout=5.0)
        if dflt != allm:
            rec["known_bad_reference_default"] = pairs(dflt)
        out.append(rec)
    errs = []
    for rx in PARSE_ERRORS:
        r = ref.call("all", rx.encode("latin1"), b"abc")
        assert r == -1, (rx, r)
        errs.append(dict(regex=rx, status="ParserError"))
    for rx in REJECTED:
        errs.append(dict(regex=rx, status="rejected"))
    return dict(vectors=out, errors=errs)


# --------------------------------------------------------------------------- fuzz
ALPHABETS = ["ab", "abc", "ab\n", "abcx_", "01a ", "a\r\nb"]


class RegexGen:
    """Random patterns from the language the reference accepts without crashing."""

    def __init__(self, rng, alphabet):
        self.r = rng
        self.a = alphabet

    def lit_char(self):
        c = self.r.choice(self.a)
        return {"\n": "\\n", "\t": "\\t", "\r": "\r"}.get(c, c)

    def atom(self, depth):
        k = self.r.random()
        if k < 0.45:
            return "".join(self.lit_char() for _ in range(self.r.randint(1, 4)))
        if k < 0.55:
            return "."
        if k < 0.68:
            body = "".join(c for c in self.r.sample(self.a, self.r.randint(1, min(3, len(self.a))))
                           if c not in "\n\r\t")
            if not body:
                body = "a"
            if self.r.random() < 0.3 and len(body) >= 2:
                lo, hi = sorted(body[:2])
                body = f"{lo}-{hi}" + body[2:]
            return "[" + ("^" if self.r.random() < 0.25 else "") + body + "]"
        if k < 0.74:
            return self.r.choice(["^", "$"])
        if k < 0.80:
            return self.r.choice(["\\d", "\\D", "\\s", "\\S"])
        if depth <= 0:
            return self.lit_char()
        return "(" + self.alt(depth - 1) + ")"

    def quantified(self, depth):
        a = self.atom(depth)
        k = self.r.random()
        if k < 0.6:
            return a
        q = self.r.choice(["*", "+", "?", "{2}", "{1,2}", "{0,2}", "{2,}", "{,2}", "{1}", "{0,1}", "{2,3}", "{1,3}"])
        return a + q

    def concat(self, depth):
        return "".join(self.quantified(depth) for _ in range(self.r.randint(1, 3)))

    def alt(self, depth):
        return "|".join(self.concat(depth) for _ in range(self.r.choice([1, 1, 1, 2, 2, 3])))


def gen_fuzz(ref: RefProc, count=2500, seed=20260926):
    rng = random.Random(seed)
    out = []
    bad = 0
    while len(out) < count:
        alphabet = rng.choice(ALPHABETS)
        rx = RegexGen(rng, alphabet).alt(2)
        n = rng.choice([0, 1, 3, 8, 17, 33, 64, 100])
        text = "".join(rng.choice(alphabet) for _ in range(n))
        rxb, txb = rx.encode("latin1"), text.encode("latin1")
        allm = ref.call("all", rxb, txb, timeout=5.0)
        if not isinstance(allm, list):
            bad += 1
            continue
        full = ref.call("full", rxb, txb, timeout=5.0)
        if full not in (0, 1):
            bad += 1
            continue
        out.append(dict(regex=rx, text=text, ref_all=pairs(allm), ref_full=full))
    print(f"fuzz: {len(out)} vectors, {bad} skipped (reference crashed / timed out / parse error)")
    return out




# --------------------------------------------------------------------------- bytes >= 0x80
# The reference compares bracket ranges as SIGNED bytes (src/x64/codegen-x64.cc:902-907: greater_equal / less_equal on
# movsx'd characters), its `.` / `\S` / `\D` / negated brackets take any byte, and its fast-forward literals are byte
# strings: everything here holds bytes >= 0x80 in the pattern, in the text, or both.  Texts of up to 5000 bytes, so that
# the engine's window scans (nibble filter, 2-bit codes: bytes >= 0x80 alias ASCII ones there) see them, not only its
# small-text kernel.
ALPHABETS_HI = ["a\xe9\xff", "ab\x80\x7f", "\x80\x81\xfe\xff", "a\x00\xff\x01", "ab\xc3\xa9 ", "az\x7e\x7f\x80\x81", "\n\xe9a\r",
                "i\xe9\x69\x29\xa9", "\xe0\xe1\xe2abc", "0\xb0\xf09"]
HIGH_PATTERNS = [
    # signed ranges around the sign change, negations, classes of high bytes only
    "[\x7e-\x81]", "[\x7e-\x81]+", "[^\x80]", "[^\x80]+", "[\x80-\xff]", "[\x80-\xff]+", "[^\x80-\xff]+", "[a-\xff]", "[a-\xff]+x",
    "[\xf0-\x10]", "[\xf0-\x10]+", "[\x01-\x7f]+", "[\x80-\x8f\xe0-\xef]+", "[^a\xe9]", "[\xe9\xff]+a", "a[\x80-\xff]b", "[\xe0-\xe2]{2,3}",
    "[\xfe-\xff][\x80-\x81]", "[^\xff]\xff", "[\x7f\x80]{2}",
    # . \S \D \s \d over Latin-1 / UTF-8 text
    ".", ".+", "\\S+", "\\D+", "\\S\\D", "a.b", "\xe9.\xe9", ".\xff", "\\D\xe9", "\\S{2}\xe9", "^.\xe9", "\xe9$", "^\xff+$", "(\xe9|\xff)+", "\xc3\xa9",
    "\xc3[\x80-\xbf]", "(\xc3\xa9)+", "caf\xe9", "na\xefve|\xfcber", "\xe9t\xe9", "[a-z]*\xe9[a-z]*",
    # literal windows of high bytes: 4, 5 and 8 bytes (window filters), alternations of them, with a prefix / suffix
    "\x80\x81\x82\x83", "\xe0\xe1\xe2\xe0\xe1", "\x80\x81\x82\x83\x84\x85\x86\x87", "\xff\xfe\xfd\xfc\xfb\xfa\xf9\xf8", "\xe9\xe9\xe9\xe9\xe9\xe9",
    "\x80\x81\x82\x83|\xff\xfe\xfd\xfc", "a\xe0\xe1\xe2\xe0|b\xe1\xe2\xe0\xe1", "[ab]\x80\x81\x82\x83", "\x80\x81\x82\x83[ab]", "x+\xe9\xe9\xe9\xe9", "\xe9\xe9\xe9\xe9x*",
    "(\x80\x81){2,3}", "\xa9\x29\x69\xe9", "i\xe9i\xe9i\xe9i\xe9", "\xf0\xb0\xf0\xb0|0909",
    # \xHH escapes (hex letters are mis-decoded by the reference: kept, whatever it answers)
    "\\x80", "\\xe9+", "\\xff", "[\\x80-\\x8f]",
]


def gen_highbyte(ref: RefProc, seed=20260928):
    rng = random.Random(seed)
    out = []
    bad = 0

    def text_of(alphabet, n, plant=()):
        t = [rng.choice(alphabet) for _ in range(n)]
        for p in plant:
            if n > len(p):
                at = rng.randrange(0, n - len(p) + 1)
                t[at:at + len(p)] = list(p)
        return "".join(t)

    def add(rx, text):
        nonlocal bad
        rxb, txb = rx.encode("latin1"), text.encode("latin1")
        allm = ref.call("all", rxb, txb, timeout=10.0)
        if not isinstance(allm, list):
            bad += 1
            return
        full = ref.call("full", rxb, txb, timeout=10.0)
        if full not in (0, 1):
            bad += 1
            return
        # (texts as base64: a high byte costs six characters as a JSON escape; long match lists as count + digest)
        v = dict(regex=rx, text_b64=base64.b64encode(txb).decode("ascii"), ref_full=full)
        if len(allm) > 100:
            v["ref_digest"] = dict(count=len(allm), sha256=hashlib.sha256(repr([tuple(p) for p in allm]).encode()).hexdigest())
        else:
            v["ref_all"] = pairs(allm)
        out.append(v)

    every_byte = "".join(chr(c) for c in range(256))
    for rx in HIGH_PATTERNS:
        # what a literal of the pattern looks like, planted into the random texts (patterns of literals and classes)
        lits = [m for m in re.findall(r"(?:[^\\\[\]()|*+?{}.^$]){2,}", rx)]
        add(rx, every_byte)
        add(rx, every_byte[::-1] * 2)
        for k, alphabet in enumerate(rng.sample(ALPHABETS_HI, 3)):
            for n in (0, 1, 9, 33, 200) + ((1100,) if k < 2 else (5000,)):
                add(rx, text_of(alphabet, n, plant=[rng.choice(lits) for _ in range(n // 40)] if lits else ()))
    # random patterns over the high alphabets (the generator of gen_fuzz; its brackets then hold high bytes and signed ranges)
    want = len(out) + 700
    while len(out) < want:
        alphabet = rng.choice(ALPHABETS_HI)
        gen_alphabet = alphabet.replace("\x00", "")      # (a C string: no NUL in the pattern)
        rx = RegexGen(rng, gen_alphabet).alt(2)
        n = rng.choice([0, 1, 3, 8, 17, 33, 64, 100, 300, 1000])
        add(rx, text_of(alphabet, n))
    with_high = sum(1 for v in out if any(ord(c) >= 0x80 for c in v["regex"]) or any(c >= 0x80 for c in base64.b64decode(v["text_b64"])))
    print(f"highbyte: {len(out)} vectors ({with_high} with a byte >= 0x80), {bad} skipped (reference crashed / timed out / parse error)")
    return out


# --------------------------------------------------------------------------- the ring artefact
ARTEFACT_PATTERNS = [".{0,2}.", "[a-f]+[0-9][a-f]", "(ab|ba)+", "[xy]+z[xy]", "x+yx", "(aa|aaa)+", "[a-z]+@[a-z]+", "^.{0,2}.",
                     "[ab]+b[ab]", "a+(b|ca)", ".{1,3}b", "(a|bc){1,3}d?", "[a-c]+d[a-c]*"]


def gen_artefact(ref: RefProc, seed=20260927):
    from checkers import Oracle
    oracle = Oracle()
    rng = random.Random(seed)
    out, differ = [], 0
    pats = [(p, "abcdef0123xyz@ \n") for p in ARTEFACT_PATTERNS]
    tries = 0
    while len(pats) < 60 and tries < 20000:     # random patterns on which the artefact shows
        tries += 1
        alphabet = rng.choice(ALPHABETS)
        rx = RegexGen(rng, alphabet).alt(2)
        text = "".join(rng.choice(alphabet) for _ in range(120)).encode("latin1")
        w = oracle.match_all(rx.encode("latin1"), text)
        if isinstance(w, list) and w != oracle.match_all_spec(rx.encode("latin1"), text):
            pats.append((rx, alphabet))
    for rx, alphabet in pats:
        rxb = rx.encode("latin1")
        for n in (5, 12, 40, 90, 200, 350, 600):
            for alpha in (alphabet, alphabet[:max(2, len(alphabet) // 3)]):
                text = "".join(rng.choice(alpha) for _ in range(n))
                txb = text.encode("latin1")
                allm = ref.call("all", rxb, txb, timeout=5.0)
                if not isinstance(allm, list):
                    continue
                full = ref.call("full", rxb, txb, timeout=5.0)
                if full not in (0, 1):
                    continue
                differ += pairs(allm) != [list(x) for x in oracle.match_all_spec(rxb, txb)]
                out.append(dict(regex=rx, text=text, ref_all=pairs(allm), ref_full=full))
    print(f"artefact: {len(out)} vectors over {len(pats)} patterns; the reference differs from the documented semantics on {differ}")
    return out

# --------------------------------------------------------------------------- bench-shaped
def digest(ms):
    h = hashlib.sha256()
    for b, e in ms:
        h.update(int(b).to_bytes(8, "little"))
        h.update(int(e).to_bytes(8, "little"))
    return h.hexdigest()


def gen_bench(ref: RefProc):
    out = {"bench": [], "regexdna": {}}
    n = 1 << 16
    for i, (rx, lo, hi) in enumerate(W.BENCH_REGEXES):
        seed = 1000 + i
        text = W.random_ascii_numpy(n, seed, ord(lo), ord(hi))
        rng = random.Random(seed)
        # plant a few strings from the pattern's language + decoys
        plants = {
            0: [b"abcdefgh"], 1: [b"abcdefgh"], 2: [b"abcdefgh"],
            3: [W.complex_regex_sample(rng) for _ in range(6)] + [b"abcdefgh"] * 4,
            4: [b"alternation", b"strings"], 5: [b"alternation", b"more", b"than", b"two", b"different", b"strings"],
            6: [b"rather_long_string", b"min"],
            7: [b"complexregexpalternation", b"xcregexpalternation", b"stringsat", b"stringsthe", b"stringsdas well"],
            8: [b"prefix abcd", b"prefix 1234", b"prefix 12"], 9: [b"abcd suffix", b"1234 suffix", b"34 suffix"],
            10: [b"abcdefgh anywhere xyz", b"01 anywhere 56789", b" anywhere "],
            11: [b"some bla root blah ", b"sotherregexps bla root blah ", b"f abcdefgh boottt xyz ", b"some 00 foot 5678",
                 b"u 00 foot 5678"],
        }[i]
        offs = W.plant_offsets(n, 40, 24, seed, boundaries=[16, 1024, 4096, 16384, 32768])
        for k, o in enumerate(offs):
            p = plants[k % len(plants)]
            text[o:o + len(p)] = W.np.frombuffer(p, dtype=W.np.uint8)
        tb = text.tobytes()
        allm = ref.call("all", rx.encode(), tb, timeout=120.0)
        assert isinstance(allm, list), (rx, allm)
        out["bench"].append(dict(regex=rx, low=lo, high=hi, seed=seed, n=n, plant_offsets=offs,
                                 plants=[s(p) for p in plants], text_sha256=hashlib.sha256(tb).hexdigest(),
                                 ref_all=pairs(allm)))
        print(f"bench[{i}] {rx!r}: {len(allm)} matches")
    for nf in (1000, 50000):
        raw = W.fasta_raw_numpy(nf).tobytes()
        stripped = W.fasta_stripped_numpy(nf).tobytes()
        entry = dict(raw_size=len(raw), stripped_size=len(stripped),
                     raw_sha256=hashlib.sha256(raw).hexdigest(),
                     stripped_sha256=hashlib.sha256(stripped).hexdigest(), patterns=[])
        strip = ref.call("all", W.REGEXDNA_STRIP.encode(), raw, timeout=300.0)
        entry["strip"] = dict(regex=W.REGEXDNA_STRIP, count=len(strip), digest=digest(strip))
        for rx in W.REGEXDNA_PATTERNS:
            allm = ref.call("all", rx.encode(), stripped, timeout=300.0)
            assert isinstance(allm, list)
            rec = dict(regex=rx, count=len(allm), digest=digest(allm))
            dflt = ref.call("all", rx.encode(), stripped, flags=RefProc.DEFAULT, timeout=300.0)
            if dflt != allm:
                rec["known_bad_reference_default_count"] = len(dflt) if isinstance(dflt, list) else dflt
            entry["patterns"].append(rec)
        # the 11 IUB replacements, sequentially as sample/regexdna.cc:83-85 does
        text = stripped
        for code, repl in W.REGEXDNA_IUB:
            ms = ref.call("all", code.encode(), text, timeout=300.0)
            outb, p = bytearray(), 0
            for b, e in ms:
                outb += text[p:b] + repl.encode()
                p = e
            outb += text[p:]
            text = bytes(outb)
        entry["replaced_size"] = len(text)
        entry["replaced_sha256"] = hashlib.sha256(text).hexdigest()
        out["regexdna"][str(nf)] = entry
        print(f"regexdna n={nf}: counts {[p['count'] for p in entry['patterns']]} sizes "
              f"{entry['raw_size']}/{entry['stripped_size']}/{entry['replaced_size']}")
    return out


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"), ensure_ascii=True)
    print(f"wrote {name}: {os.path.getsize(path)} bytes")


def main():
    ref = RefProc()
    which = sys.argv[1:] or ["testcc", "semantics", "fuzz", "artefact", "highbyte", "bench"]
    if "testcc" in which:
        dump("testcc_vectors.json", gen_testcc(ref))
    if "semantics" in which:
        dump("semantics_vectors.json", gen_semantics(ref))
    if "fuzz" in which:
        dump("fuzz_vectors.json", gen_fuzz(ref))
    if "artefact" in which:
        dump("artefact_vectors.json", gen_artefact(ref))
    if "highbyte" in which:
        dump("highbyte_vectors.json", gen_highbyte(ref))
    if "bench" in which:
        dump("bench_vectors.json", gen_bench(ref))


if __name__ == "__main__":
    main()
