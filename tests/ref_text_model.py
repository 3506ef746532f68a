"""A third statement of the TEXT layer of `fasta demultiplex` — plain Python, written from the reference's Rust lines and the Rust
standard library's documented behaviour, NOT from this repo's C++ host (seqkit_amd/csrc/fasta_main.cpp) or C oracle
(oracle/fasta_oracle_main.c): those two come from one pen, and a shared misreading of `str::trim_end`, of the regex's
leftmost-first match, of `{:.1}` or of `chars().zip()` passes every comparison between them ("what's weak" 1).

What is modelled (src/fasta_demultiplex.rs, line numbers of the reference):
  :58-104   the sample sheet: `#` lines skipped, `line.trim().split('\\t')`, fewer than two columns skipped, empty barcode, unequal
            lengths (BYTES: `str::len`), duplicate names;
  :112-150  per cluster: header must start with `@`; barcode from the index files (`line.trim_end()`, joined with `+`) or from the
            header's first match of ` BC:[ACGTNacgtn+]+` (Rust `regex`: leftmost-first, the class is ASCII), drained from the header;
            its length in BYTES against the sheet's;
  :152-194  `barcode_diff` over BYTES with `N` / `U` of the sheet as wildcards, first and last argmin, the decision, the warning;
  :196-238  the UMI: `sample.barcode.chars().zip(barcode.chars())` — CHARS, not bytes —, `header.trim_end()`, the mate's header;
  :263-264  the summary with `{:.1}`.
`read_line` needs valid UTF-8 (src/common.rs:104-110: anything else is "I/O error while reading from file.").
Rust `char::is_whitespace` / `str::trim*` use the Unicode White_Space property (std docs): U+0009..U+000D, U+0020, U+0085, U+00A0,
U+1680, U+2000..U+200A, U+2028, U+2029, U+202F, U+205F, U+3000 — not U+001C..U+001F, which Python's str.strip() removes."""
import re

WHITE_SPACE = frozenset([0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, *range(0x2000, 0x200B), 0x2028, 0x2029, 0x202F, 0x205F, 0x3000])
BC_RE = re.compile(r" BC:[ACGTNacgtn+]+")          # Python's re is leftmost-first (backtracking) like Rust's regex for this pattern: greedy class, no alternation


def trim_end(s: str) -> str:
    n = len(s)
    while n and ord(s[n - 1]) in WHITE_SPACE:
        n -= 1
    return s[:n]


def trim(s: str) -> str:
    s = trim_end(s)
    k = 0
    while k < len(s) and ord(s[k]) in WHITE_SPACE:
        k += 1
    return s[k:]


def fmt1(x: float) -> str:
    """Rust `{:.1}` of an f64: correctly rounded from the exact binary value (ties to even on that value), `NaN`, `inf`."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    return f"{x:.1f}"


class Exit(Exception):
    def __init__(self, code):
        self.code = code


class Lines:
    """FileReader::read_line over bytes: (ok, line) — a line is text up to and including '\\n'; invalid UTF-8 is the I/O error"""

    def __init__(self, data: bytes, err):
        self.data, self.at, self.err = data, 0, err

    def read_line(self):
        if self.at >= len(self.data):
            return False, ""
        e = self.data.find(b"\n", self.at)
        e = len(self.data) if e < 0 else e + 1
        raw = self.data[self.at:e]
        self.at = e
        try:
            return True, raw.decode("utf-8")
        except UnicodeDecodeError:
            self.err("ERROR: I/O error while reading from file.\n")
            raise Exit(255)


def barcode_diff(obs: bytes, cand: bytes) -> int:
    assert len(obs) == len(cand)
    return sum(1 for o, c in zip(obs, cand) if c not in b"NU" and o != c)


def demultiplex(sheet: bytes, fq1: bytes, fq2=None, index=(), dry_run=0, panic_below_100=True):
    """-> (exit code, stdout, stderr, {file name: decompressed content}).  index: the --index1 / --index2 files' bytes.
    panic_below_100: the reference's `&entries[0..100]` (:258) panics when the dry run's table has fewer than a hundred entries — the
    oracle command line does the same; the HIP host prints the entries there are: False models that."""
    out, err, files = [], [], {}
    e = err.append
    try:
        fastq = [Lines(fq1, e)] + ([Lines(fq2, e)] if fq2 is not None else [])
        paired = len(fastq) == 2
        index_fastq = [Lines(b, e) for b in index]
        e("Reading sample sheet...\n")
        samples = []                                           # [name, barcode, total]
        barcode_len = 0
        sh = Lines(sheet, e)
        while True:
            ok, line = sh.read_line()
            if not ok:
                break
            if line.startswith("#"):
                continue
            cols = trim(line).split("\t")
            if len(cols) < 2:
                continue
            name = cols[0]
            if cols[1] == "":
                e(f"ERROR: Sample {name} has no barcode.\n")
                raise Exit(255)
            blen = len(cols[1].encode())
            if barcode_len == 0:
                barcode_len = blen
            elif blen != barcode_len:
                e("ERROR: Barcodes in sample sheet must all be of same length.\n")
                raise Exit(255)
            if dry_run == 0:
                for fn in ([f"{name}_1.fq.gz", f"{name}_2.fq.gz"] if paired else [f"{name}.fq.gz"]):
                    files[fn] = []                            # (a second sample of the same name truncates the file again: the name check comes later)
            samples.append([name, cols[1], 0])
        for s in range(len(samples)):
            for k in range(s + 1, len(samples)):
                if samples[s][0] == samples[k][0]:
                    e(f"ERROR: Sample {samples[s][0]} is listed multiple times in sample sheet.\n")
                    raise Exit(255)
        e(f"Starting demultiplexing in {'paired' if paired else 'single'} end mode...\n")
        total = identified = 0
        extra = {}
        while True:
            ok, header = fastq[0].read_line()
            if not ok:
                break
            if not header.startswith("@"):
                e(f"ERROR: Invalid FASTQ header line:\n{header}\n")
                raise Exit(255)
            barcode = ""
            if index_fastq:
                for ifq in index_fastq:
                    if barcode:
                        barcode += "+"
                    _, line = ifq.read_line()
                    if not line.startswith("@"):
                        raise Exit(101)                       # assert!
                    _, line = ifq.read_line()
                    barcode += trim_end(line)
                    _, line = ifq.read_line()
                    if not line.startswith("+"):
                        raise Exit(101)
                    ifq.read_line()
            else:
                m = BC_RE.search(header)
                if m is None:
                    e("ERROR: No BC:xxxx field found.\n")
                    raise Exit(255)
                barcode += header[m.start() + 4:m.end()]
                header = header[:m.start()] + header[m.end():]
            blen = len(barcode.encode())
            if blen != barcode_len:
                e(f"ERROR: Sequenced barcode {barcode} is of different length ({blen} nt) than barcodes in the sample sheet ({barcode_len} nt).\n")
                raise Exit(255)
            best = last = 0
            lowest = None
            for s, (_, bc, _) in enumerate(samples):
                d = barcode_diff(barcode.encode(), bc.encode())
                if lowest is None or d < lowest:
                    lowest, best, last = d, s, s
                elif d == lowest:
                    last = s
            total += 1
            write = False
            if lowest is not None and lowest <= 1:
                if best == last:
                    identified += 1
                    samples[best][2] += 1
                    write = not dry_run > 0
                else:
                    e(f"WARNING: Sequenced barcode {barcode} was an equally good match ({lowest} mismatches) for samples {samples[best][0]} ({samples[best][1]}) and "
                      f"{samples[last][0]} ({samples[last][1]}), and was therefore not assigned to any sample.\n")
            elif dry_run > 0:
                extra[barcode] = extra.get(barcode, 0) + 1
            if write:
                name, sbc, _ = samples[best]
                umi = "".join(o for c, o in zip(sbc, barcode) if c == "U")       # chars().zip(chars())
                w = files[f"{name}_1.fq.gz" if paired else f"{name}.fq.gz"]
                w.append(trim_end(header) + (f" UMI:{umi}" if umi else "") + "\n")
                for _ in range(3):
                    w.append(fastq[0].read_line()[1])
                if paired:
                    _, line = fastq[1].read_line()
                    if not index_fastq:
                        m = BC_RE.search(line)
                        if m is not None and m.end() > 0:
                            line = line[:m.start()] + line[m.end():]
                    w2 = files[f"{name}_2.fq.gz"]
                    w2.append(trim_end(line) + (f" UMI:{umi}" if umi else "") + "\n")
                    for _ in range(3):
                        w2.append(fastq[1].read_line()[1])
            else:
                for _ in range(3):
                    fastq[0].read_line()
                if paired:
                    for _ in range(4):
                        fastq[1].read_line()
            if dry_run > 0 and total >= dry_run:
                break
        if dry_run > 0:
            e(f"Dry run completed with {total} clusters. Barcodes found:\n")
            entries = [(n, t) for n, _, t in samples] + list(extra.items())
            if panic_below_100 and len(entries) < 100:
                raise Exit(101)
            out.append(entries)                               # (the caller compares as the reference's order allows: by count, ties in any order)
        pct = (identified / total * 100.0) if total else float("nan")
        e(f"{identified} / {total} ({fmt1(pct)}%) clusters carried a barcode matching one of the provided samples.\n")
        code = 0
    except Exit as x:
        code = x.code
    return code, out, "".join(err), {k: "".join(v) for k, v in files.items()}
