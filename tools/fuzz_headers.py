#!/usr/bin/env python3
"""CPU fuzz of the host/device algorithm headers through the test driver tests/support/libcarry_exec.so
(built by the CPU tests; run `python -m pytest tests/test_carry_scan.py -q -k behind` once if it is missing):

  exact  exact_replay.h against the oracle: whole text and random 3-way splits into ranges, random chunk sizes
         (round 2: 96 000 cases, 0 mismatches)
  swar   dense_swar.h against the scalar automaton, rj_lane_longest_short against rj_lane_longest, random patterns
         (round 2: 6600 + 6500 plans, 0 mismatches)

  lds    lds_walk.h against the walkers it replaces (device_program.h, behind_walk.h), random patterns

  carry  carry_scan.h against the oracle's documented semantics, whole texts and two ranges with the carry over a cut

usage: fuzz_headers.py exact|swar|lds|carry [seed] [cases]"""
import ctypes, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
SO = os.path.join(ROOT, "tests", "support", "libcarry_exec.so")
mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
sys.argv = [sys.argv[0]] + (sys.argv[2:] + ["1", "5000"])[:2]


def fuzz_exact():
    from checkers import Oracle
    from make_golden import RegexGen, ALPHABETS
    o=Oracle()
    lib=ctypes.CDLL(SO)
    _u64p=ctypes.POINTER(ctypes.c_uint64)
    lib.ce_exact_range.restype=ctypes.c_long
    lib.ce_exact_range.argtypes=[ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _u64p, ctypes.c_uint64, _u64p, _u64p, ctypes.POINTER(ctypes.c_int)]
    def exact(rx,tx,chunk,sb=0,se=None):
        se=len(tx)+1 if se is None else se
        cap=len(tx)+2; buf=(ctypes.c_uint64*(2*cap))(); a=ctypes.c_uint64(); b=ctypes.c_uint64(); r=ctypes.c_int()
        n=lib.ce_exact_range(rx,tx,len(tx),chunk,sb,se,buf,cap,ctypes.byref(a),ctypes.byref(b),ctypes.byref(r))
        if n<0: return int(n)
        return [(int(buf[2*i]),int(buf[2*i+1])) for i in range(n)]
    seed=int(sys.argv[1]); N=int(sys.argv[2]); rng=random.Random(seed)
    ALPH=ALPHABETS+["ab\n\r","xyz ^$","aA0-"]
    n=bad=0; t0=time.time()
    for it in range(N):
        alphabet=rng.choice(ALPH)
        rx=RegexGen(rng, alphabet).alt(3).encode('latin1')
        if o.status(rx)!=0: continue
        tx="".join(rng.choice(alphabet) for _ in range(rng.choice([0,1,9,33,120,500]))).encode('latin1')
        want=o.match_all(rx,tx)
        if isinstance(want,int): continue
        chunk=rng.choice([1,2,5,16,64,1024])
        got=exact(rx,tx,chunk)
        if got==-9: continue
        n+=1
        if got!=want:
            print("MISMATCH whole", rx, tx, chunk, got[:4], want[:4]); bad+=1
        k=rng.randrange(0,len(tx)+2); k2=rng.randrange(k,len(tx)+2)
        parts=[]
        for lo,hi in ((0,k),(k,k2),(k2,len(tx)+1)):
            if lo<hi:
                g=exact(rx,tx,rng.choice([3,16,256]),lo,hi); parts+=g
        if parts!=want:
            print("MISMATCH split", rx, tx, (k,k2), parts[:4], want[:4]); bad+=1
        if bad>5: break
    print(f"seed {seed}: checked {n}, mismatches {bad}, {time.time()-t0:.0f}s")



def fuzz_swar():
    from make_golden import RegexGen, ALPHABETS
    lib=ctypes.CDLL(SO)
    _u64p=ctypes.POINTER(ctypes.c_uint64)
    lib.ce_swar_check.restype=ctypes.c_long
    lib.ce_swar_check.argtypes=[ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, _u64p]
    lib.ce_short_check.restype=ctypes.c_long
    lib.ce_short_check.argtypes=[ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, _u64p]
    seed=int(sys.argv[1]); N=int(sys.argv[2]); rng=random.Random(seed)
    ALPH=ALPHABETS+["abcdef0123456789","xyz@#AZ az","\x80\x90\xffab"]
    used=short=bad=0; t0=time.time()
    for it in range(N):
        alphabet=rng.choice(ALPH)
        rx=RegexGen(rng, alphabet).alt(2).encode('latin1')
        tx=bytes(rng.choice(alphabet.encode('latin1')) for _ in range(16*40+4))
        st=(ctypes.c_uint64*4)()
        r=lib.ce_swar_check(rx,tx,len(tx),st)
        if r>=0:
            used+=1
            if r!=0: print("SWAR MISMATCH", rx, r); bad+=1
        c=ctypes.c_uint64(0)
        r=lib.ce_short_check(rx,tx[:200],200,ctypes.byref(c))
        if r>=0:
            short+=1
            if r!=0: print("SHORT MISMATCH", rx, r); bad+=1
        if bad>5: break
    print(f"seed {seed}: swar plans {used}, short plans {short}, mismatches {bad}, {time.time()-t0:.0f}s")



def fuzz_lds():
    """lds_walk.h (what verify_lds.hip walks in LDS: padded tables, rows by position, text through block / lane / wave
    windows with guard bytes) against device_program.h / behind_walk.h on random patterns: longest match from every
    start, forward check and backward walk from every (text position, automaton position), behind candidates."""
    from checkers import Oracle
    from make_golden import RegexGen, ALPHABETS
    o = Oracle()
    lib = ctypes.CDLL(SO)
    lib.ce_lds_walk_check.restype = ctypes.c_long
    lib.ce_lds_walk_check.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    seed = int(sys.argv[1]); N = int(sys.argv[2]); rng = random.Random(seed)
    ALPH = ALPHABETS + ["ab\n\r", "abcdefgh12 \n"]
    used = bad = 0; total = 0; t0 = time.time()
    for it in range(N):
        alphabet = rng.choice(ALPH)
        rx = RegexGen(rng, alphabet).alt(3)
        if rng.random() < 0.4:
            rx = rng.choice(["", ".*", "[a-z]+", "^"]) + rx + rng.choice(["", "{2,5}", "+", "abc", "$"])
        rx = rx.encode("latin1")
        if b"\0" in rx or o.status(rx) != 0: continue
        tx = "".join(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 17, 48, 130]))).encode("latin1")
        for walk in (1 << 20, rng.choice([3, 7, 19])):
            c = ctypes.c_uint64(0)
            r = lib.ce_lds_walk_check(rx, tx, len(tx), walk, ctypes.byref(c))
            if r < 0: break           # (the pattern does not take these walkers: > 4 words, or no tables)
            used += 1; total += c.value
            if r != 0:
                print("LDS WALK MISMATCH", rx, tx, walk, r); bad += 1
        if bad > 5: break
    print(f"seed {seed}: lds-walk runs {used}, comparisons {total}, mismatches {bad}, {time.time()-t0:.0f}s")


def fuzz_carry():
    """carry_scan.h (the linear-time matcher) against the oracle's documented semantics on random patterns: whole texts
    with random sub-chunk sizes, and two ranges with the selection state carried over a random cut."""
    from checkers import Oracle
    from make_golden import RegexGen, ALPHABETS
    o = Oracle()
    lib = ctypes.CDLL(SO)
    _u64p = ctypes.POINTER(ctypes.c_uint64)
    lib.ce_match_range.restype = ctypes.c_long
    lib.ce_match_range.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                   ctypes.c_uint64, ctypes.c_int, _u64p, ctypes.c_uint64]
    def carry(rx, tx, sub, sb=0, se=None, cur=0, prev_end=0, have=False):
        se = len(tx) + 1 if se is None else se
        cap = len(tx) + 2; buf = (ctypes.c_uint64 * (2 * cap))()
        n = lib.ce_match_range(rx, tx, len(tx), sub, sb, se, cur, prev_end, int(have), buf, cap)
        return int(n) if n < 0 else [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]
    seed = int(sys.argv[1]); N = int(sys.argv[2]); rng = random.Random(seed)
    ALPH = ALPHABETS + ["ab\n\r", "xyz ^$", "acgt"]
    used = bad = 0; t0 = time.time()
    for it in range(N):
        alphabet = rng.choice(ALPH)
        rx = RegexGen(rng, alphabet).alt(3)
        if rng.random() < 0.3:
            rx = rng.choice(["", ".*", "[a-z]+"]) + rx + rng.choice(["", "+", "*", "{2,}"])
        rx = rx.encode("latin1")
        if b"\0" in rx or o.status(rx) != 0: continue
        tx = "".join(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 9, 70, 300, 900]))).encode("latin1")
        spec = o.match_all_spec(rx, tx)
        if isinstance(spec, int): continue
        got = carry(rx, tx, rng.choice([1, 3, 16, 64, 4096]))
        if isinstance(got, int): continue      # (wider than the driver's limit)
        used += 1
        ok = got == spec
        if ok and len(tx) > 1:
            cut = rng.randrange(1, len(tx) + 1)
            first = carry(rx, tx, 32, 0, cut)
            state = (0, 0, False)
            if first:
                b, e = first[-1]; state = (e if e > b else b + 1, e, True)
            ok = first + carry(rx, tx, 32, cut, len(tx) + 1, *state) == spec
        if not ok:
            print("CARRY MISMATCH", rx, tx[:60]); bad += 1
            if bad > 5: break
    print(f"seed {seed}: carry-scan cases {used}, mismatches {bad}, {time.time()-t0:.0f}s")


{"exact": fuzz_exact, "swar": fuzz_swar, "lds": fuzz_lds, "carry": fuzz_carry}[mode]()
