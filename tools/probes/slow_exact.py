"""A pathological set found by tools/fuzz_reuse.py 40 10701 (round 6, last session): its third pattern -- eight nested groups with `^` / `$`
inside repetitions, at risk of the reference's ring artefact, an automaton too wide for exact_replay.hip -- takes the one-lane
exact_sequential kernel: 18.5 s per 100 KB of `ab\\n` text (exact_path 1), minutes per MiB.  Not a hang; not new (the same with
RJ_NO_RUNS=1 RJ_NO_PAIRS=1 RJ_NO_WINDOW_RUNS=1).   python tools/probes/slow_exact.py [pattern index, -1 = all] [bytes]"""
import faulthandler, sys, os, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, rejit_amd
pats = [b'.(aa[^a]){,2}aa\n{,2}|\n\n\n\n'.replace(b"\n", b"\\n"),
        b'(\\na\\n\\n.{1}|.{2,}(\\S{2,3}$+bab|.b*b)[ba])\\nb\\n\\n{2}(\\na|a\\n([ab]{,2}b\\nba.|[a-b]){,2}[^a])|^{2,}.\\na|\\d',
        b'[a]+(^[ba]{1,2}.|[a-b]?\\Db\\n*|\\nb\\nb+){,2}^{1,2}|(\\n{1,2}([a]|b\\nbaa{0,2}b\\n{0,2})|($){2,3}a{0,1}){2,}(\\na\\nb[ab]\\na{0,2})(\\dab\\nb*|([a]|aba{0,1}bb\\d)|a{2}a\\na)+|([ba]+b{0,1}){2,3}$(aa\\na){2,3}',
        b'a{,2}((aab.{2,}.{2,3}).{0,2}a)|[ba]']
rng = random.Random(5)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000000
piece = "".join(rng.choice("ab\n") for _ in range(200000))
text = (piece * (n // len(piece) + 1))[:n].encode("latin1")
d = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for i, p in enumerate(pats):
    if only >= 0 and i != only:
        continue
    sc = rejit_amd.Scan(rejit_amd.Program(p))
    faulthandler.cancel_dump_traceback_later(); faulthandler.dump_traceback_later(45, exit=True, file=sys.stderr)
    print("pattern", i, p[:60], flush=True)
    t0 = time.perf_counter()
    k = sc.run(d.data_ptr(), n)
    print("   ", k, "matches", round(time.perf_counter() - t0, 3), "s", {a: b for a, b in sc.stats().items() if b}, flush=True)
