"""Random FASTA / FASTQ files with irregular records injected at a given rate -- shared by tools/fuzz_text_route.py (GPU box: text route against host
route and the compiled reference) and tools/fuzz_host_sanitizers.py (here: the host pipeline under ASan / UBSan / TSan).  (Test infrastructure.)"""
import numpy as np


def make_file(rng, path, s, k, n, fastq, irr, mixed, hdr_style):
    """n records drawn from the genome into `path` (the file is the test case) -> the longest read length drawn"""
    L = int(rng.choice([8, 12, k + 3, 60, 100, 150, 150, 250, 400, 1200, 3000]))   # (8, 12: more than one record per 24 bytes -- pieces beyond the device's record table;
    if L >= 1200:                                                                    #  1200, 3000: records that leave the 1 KB window behind a tile of the parse launch)
        n = min(n, 12000)
    if L <= k and not (fastq or rng.random() < 0.3):
        L = k + 3
    reads, roffs = s.reads(int(rng.integers(0, 1 << 30)), n, L, 3, int(rng.integers(1, 1 << 30)))
    reads = reads.reshape(n, L)
    lens = np.full(n, L) if not mixed else rng.integers(min(L, max(1, k - 4)), L + 1, size=n)
    kinds = rng.random(n) < irr
    out = []
    if not fastq and irr and rng.random() < 0.15:
        out.append(b"text in front of the first header\nACGT\n")
    for i in range(n):
        seq = reads[i, : lens[i]].tobytes()
        if hdr_style == 0:
            h = b"r%d" % i
        elif hdr_style == 1:
            h = b"read_%d length=%d some description with spaces" % (i, lens[i])
        else:
            h = b"" if i % 97 == 0 else b"x%d" % i
        if hdr_style == 2 and i % 41 == 7:
            h = h + b" " + b"long header text " * int(rng.integers(20, 90))   # (a workgroup's stretch of the paths stream beyond its LDS buffer)
        qual = b"I" * len(seq)
        if kinds[i]:
            kind = int(rng.integers(0, 12))
            if kind == 0: seq = seq.lower()
            elif kind == 1: seq = seq[: len(seq) // 2] + b"N" + seq[len(seq) // 2 + 1:]
            elif kind == 2: seq = seq + b"\r"
            elif kind == 3: seq = b""
            elif kind == 4 and not fastq: seq = seq[: len(seq) // 2] + b"\n" + seq[len(seq) // 2:]     # multi-line sequence
            elif kind == 5 and not fastq: seq = seq + b"\n"                                           # blank line behind the record
            elif kind == 6: h = h + b" >inside>"
            elif kind == 7 and not fastq: seq = b">" + seq[1:]                                          # a sequence line that starts like a header
            elif kind == 8: seq = seq[: max(1, min(len(seq), k - int(rng.integers(0, 3))))]            # at most k bases
            elif kind == 9: seq = seq[: len(seq) // 3] + b"X" + seq[len(seq) // 3 + 1:]
            elif kind == 10 and fastq: qual = b"@" + qual[1:]
            elif kind == 11 and fastq: qual = b"+" + qual[1:]
        if fastq:
            plus = b"+" + (h if i % 5 == 0 else b"")
            out.append(b"@" + h + b"\n" + seq + b"\n" + plus + b"\n" + qual[: len(seq)] + b"\n")
        else:
            out.append(b">" + h + b"\n" + seq + b"\n")
    data = b"".join(out)
    tail = int(rng.integers(0, 6)) if irr else 0
    if tail == 1 and data.endswith(b"\n"):
        data = data[:-1]                     # the last record without its newline
    elif tail == 2 and fastq:
        data = data[: len(data) - int(rng.integers(1, 40))]   # truncated tail
    elif tail == 3 and not fastq:
        data += b">dangling header"
    with open(path, "wb") as f:
        f.write(data)
    return L
