#!/usr/bin/env python3
"""grep-like search on the GPU: counterpart of the reference's sample/jrep.cc (tree walk
:408-493, per-file MatchAll :288, `^` line table :294-313, line output :336-369) built on
rj_match_all_batch -- the files of a whole batch (default 256 MiB) are matched in ONE device pass
instead of one call (and one PCIe copy, and ~50 us of latency) per file.

    python samples/jrep_gpu.py [-H] [-n] [-r|-R] [-c] [-A n] [-B n] [-C n] [--count] PATTERN PATH...
    python -m torch.distributed.run --nproc-per-node 8 samples/jrep_gpu.py -R -H -n PATTERN PATH   # one rank per GPU

Under torch.distributed every rank walks the tree, takes its share of the files (greedy packing by
size, rejit_amd.sharding.partition_files) and rank 0 prints the gathered output (RCCL all_gather).

The reference's own jrep also runs unchanged on librejit_hip.so (oracle/_ref/jrep_hip); this one
is what a many-small-files workload (BASELINE config C5) should use.
"""
import argparse
import bisect
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

RED, END = b"\x1b[31m", b"\x1b[0m"


def walk(paths, recursive, follow):
    """File names in the order jrep visits them (ftw order is directory order; we sort for determinism)."""
    for p in paths:
        if os.path.isdir(p):
            if not recursive:
                sys.stderr.write(f"jrep_gpu: {p}: is a directory\n")
                continue
            for root, dirs, files in os.walk(p, followlinks=follow):
                dirs.sort()
                for f in sorted(files):
                    full = os.path.join(root, f)
                    if os.path.isfile(full) and (follow or not os.path.islink(full)):
                        yield full
        else:
            yield p


def batches(names, limit):
    """Lists of (name, bytes) whose sizes add up to about `limit`."""
    cur, size = [], 0
    for name in names:
        try:
            with open(name, "rb") as fh:
                data = fh.read()
        except OSError as e:
            sys.stderr.write(f"jrep_gpu: {name}: {e.strerror}\n")
            continue
        cur.append((name, data))
        size += len(data) + 1
        if size >= limit:
            yield cur
            cur, size = [], 0
    if cur:
        yield cur


def print_file(out, name, data, matches, line_starts, args):
    """Every line that holds the begin of a match, once, with optional context -- the output loop of
    jrep.cc:300-400 restated over offsets.  `line_starts` = begins of the `^` matches."""
    n = len(data)
    starts = [s for s in line_starts if s < n] or [0]     # (`^` also matches after a final newline)
    ends = starts[1:] + [n]                               # line i = data[starts[i]:ends[i]]
    n_lines = len(starts)

    def line_of(pos):
        return min(bisect.bisect_right(starts, pos) - 1, n_lines - 1)

    def head(line, sep):
        h = b""
        if args.with_filename:
            h += name.encode() + sep
        if args.line_number:
            h += str(line + 1).encode() + sep
        return h

    def text_of(first, last):
        t = data[starts[first]:ends[last]]
        return t if t.endswith((b"\n", b"\r")) else t + b"\n"

    # groups of matches that are printed together: (first line, last line, matches)
    groups = []
    for (b, e) in matches:
        if b >= n and (n == 0 or data[n - 1:n] in (b"\n", b"\r")):
            continue          # a match at the end of a file that ends in a line break: no line to show
        line = line_of(b)
        last = max(line, line_of(e - 1)) if e > b else line
        if groups and line <= groups[-1][1]:
            groups[-1][1] = max(groups[-1][1], last)
            groups[-1][2].append((b, e))
        else:
            groups.append([line, last, [(b, e)]])

    context = args.before or args.after
    printed = -1              # last line written so far
    for gi, (line, last, ms) in enumerate(groups):
        lo = max(line - args.before, printed + 1)
        if context and printed >= 0 and lo > printed + 1:
            out.write(b"--\n")
        for c in range(lo, line):
            out.write(head(c, b"-") + text_of(c, c))
        out.write(head(line, b":"))
        at = starts[line]
        for (mb, me) in ms:
            out.write(data[at:mb])
            out.write(RED + data[mb:me] + END if args.color else data[mb:me])
            at = me
        tail = data[at:ends[last]]
        out.write(tail if tail.endswith((b"\n", b"\r")) or (not tail and data[:at].endswith((b"\n", b"\r"))) else tail + b"\n")
        printed = last
        nxt = groups[gi + 1][0] if gi + 1 < len(groups) else n_lines
        for c in range(last + 1, min(last + args.after + 1, n_lines, nxt)):
            out.write(head(c, b"-") + text_of(c, c))
            printed = c


def main():
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--help", action="help")
    ap.add_argument("-H", "--with-filename", action="store_true")
    ap.add_argument("-n", "--line-number", action="store_true")
    ap.add_argument("-r", "--recursive", action="store_true")
    ap.add_argument("-R", "--dereference-recursive", action="store_true")
    ap.add_argument("-c", "--color_output", dest="color", action="store_true")
    ap.add_argument("-A", "--after-context", dest="after", type=int, default=0)
    ap.add_argument("-B", "--before-context", dest="before", type=int, default=0)
    ap.add_argument("-C", "--context", type=int, default=0)
    ap.add_argument("--count", action="store_true", help="print only `file:matches` per file with matches")
    ap.add_argument("--batch-mib", type=int, default=256)
    ap.add_argument("-o", "--output", default=None, help="write the result to this file instead of stdout (with several "
                    "ranks the launcher shares one stdout between all of them and their logs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend when launched with several ranks")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("pattern")
    ap.add_argument("paths", nargs="+")
    args = ap.parse_args()
    if args.context:
        args.after = args.before = args.context

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    cdev = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            cdev = torch.device("cuda", local)
            dist.init_process_group("nccl", device_id=cdev)
        else:
            dist.init_process_group(args.backend)
    import rejit_amd
    from rejit_amd import sharding
    rejit_amd.build()
    try:
        prog = rejit_amd.Program(args.pattern.encode())
    except rejit_amd.RejitError as e:
        sys.stderr.write(f"jrep_gpu: {e}\n")
        return 2
    sol = rejit_amd.Program(b"^")      # jrep.cc:239: the line table is a MatchAll of "^"
    import io
    out = io.BytesIO() if world > 1 else (open(args.output, "wb") if args.output else sys.stdout.buffer)
    found = False
    names = walk(args.paths, args.recursive or args.dereference_recursive, args.dereference_recursive)
    if world > 1:
        # every rank sees the same tree: the assignment needs no exchange
        names = list(names)
        sizes = []
        for nm in names:
            try:
                sizes.append(os.path.getsize(nm))
            except OSError:
                sizes.append(0)
        mine = sharding.partition_files(sizes, world)[rank]
        order = {names[i]: i for i in mine}     # a file's place in the single-rank (and the reference's) output order
        names = [names[i] for i in mine]
    records = []    # (global file index, that file's output) -- several ranks: merged in file order on rank 0
    for batch in batches(names, args.batch_mib << 20):
        texts = [d for _, d in batch]
        results = prog.match_all_batch(texts)
        hit = [i for i, r in enumerate(results) if r]
        if not hit:
            continue
        found = True
        if args.count:
            for i in hit:
                line = batch[i][0].encode() + b":" + str(len(results[i])).encode() + b"\n"
                if world > 1:
                    records.append((order[batch[i][0]], line))
                else:
                    out.write(line)
            continue
        # second pass, only over the files with matches (jrep.cc:292-295)
        lines = sol.match_all_batch([texts[i] for i in hit])
        for i, ls in zip(hit, lines):
            if world > 1:
                one = io.BytesIO()
                print_file(one, batch[i][0], texts[i], results[i], [b for b, _ in ls], args)
                records.append((order[batch[i][0]], one.getvalue()))
            else:
                print_file(out, batch[i][0], texts[i], results[i], [b for b, _ in ls], args)
    if world > 1:
        import struct
        import torch
        # every record travels with its file index; rank 0 puts the files back into the walk's order, so the
        # output does not depend on the number of ranks
        blob = b"".join(struct.pack("<QQ", gi, len(o)) + o for gi, o in records)
        whole = sharding.gather_bytes(blob, rank, world, dist, device=cdev)
        if rank == 0:
            recs, at = [], 0
            while at < len(whole):
                gi, ln = struct.unpack_from("<QQ", whole, at)
                recs.append((gi, whole[at + 16:at + 16 + ln]))
                at += 16 + ln
            whole = b"".join(o for _, o in sorted(recs, key=lambda r: r[0]))
        flag = torch.tensor([int(found)], dtype=torch.int64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(flag)
        found = bool(flag.item())
        if rank == 0:
            if args.output:
                with open(args.output, "wb") as fh:
                    fh.write(whole)
            else:
                sys.stdout.buffer.write(whole)
                sys.stdout.buffer.flush()
        dist.destroy_process_group()
    else:
        out.flush()
    return 0 if found else 1


if __name__ == "__main__":
    sys.exit(main())
