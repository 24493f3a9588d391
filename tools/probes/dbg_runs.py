import sys; sys.path.insert(0,'/root/repo')
import torch, rejit_amd, random
rng = random.Random(1)
for rx in (b"a.*b", b"<[^>]*>", b"[acgt]+"):
    p = rejit_amd.Program(rx)
    print(rx, p.info())
    data = bytes(rng.choice(b"abcdefgh<>") for _ in range(70000))
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    sc = rejit_amd.Scan(p)
    print(sc.run_tensor(t), sc.stats())
