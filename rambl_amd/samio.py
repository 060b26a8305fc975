"""Files of a StrainCall run: the gene FASTA and its index, and the mapping file.

The reference shells out to samtools 0.1.19 four times per window
(/root/reference/StrainCall/StrainCall.cpp:167 faidx, :496 view, :696 mpileup,
:229/:256 faidx for the index) and parses the text.  Here the FASTA is read as the plain
text it is, and the mapping file (SAM text or BAM) is read once by the library
(rambl_amd/csrc/sc_ingest.cpp: BGZF + the BAM record layout of the SAM specification, or SAM
text) -- no subprocess, no temp files.  Parity at the samtools boundary itself stays unpinned
(SURVEY.md section 8(c)); the emulation of the tool's TEXT that the tests compare the reader
with is tests/py_ingest_mirror.py.
"""


def is_bam(path):
    with open(path, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


def parse_region(reg):
    if ":" in reg:
        name, span = reg.rsplit(":", 1)
        a, b = span.split("-")
        return name, int(a), int(b)
    return reg, None, None


class Fasta:
    """Whole-file FASTA reader (seed-gene files are a few MB)."""

    def __init__(self, path):
        self.path = path
        self.seqs = {}
        self.order = []
        name = None
        chunks = None
        with open(path) as f:
            for line in f:
                line = line.rstrip("\r\n")
                if line.startswith(">"):
                    if name is not None:
                        self.seqs[name] = "".join(chunks)
                    name = line[1:].split()[0] if len(line) > 1 else ""
                    chunks = []
                    self.order.append(name)
                elif name is not None:
                    chunks.append(line)
        if name is not None:
            self.seqs[name] = "".join(chunks)

    def fetch(self, region):
        """Sequence text `samtools faidx fa region` would print (header removed)."""
        name, a, b = parse_region(region)
        if name not in self.seqs and region in self.seqs:
            name, a, b = region, None, None
        s = self.seqs.get(name, "")
        if a is not None:
            s = s[max(0, a - 1):b]
        return s


def read_fai(path):
    """[(name, length)] from <fasta>.fai: fields 1-2 (StrainCall.cpp:233-270)."""
    out = []
    with open(path) as f:
        for line in f:
            fld = line.split()
            if fld:
                out.append((fld[0], fld[1] if len(fld) > 1 else ""))
    return out


class Alignments:
    """The mapping file (SAM text or BAM) of a StrainCall run, read and indexed once by the library (capi.NativeAln:
    sc_aln_*): what the reference gets from `samtools view` / `samtools mpileup` per window (StrainCall.cpp:496,696),
    without the subprocess and without the text.  A window's reads never become Python objects."""

    def __init__(self, path, only=None):
        """only: reference names whose records the reader keeps ([]: none, the per-reference statistics only;
        None: all) -- a rank of a multi-GPU run prices every region, then keeps its own shard."""
        from . import capi
        self.path = path
        self.native = capi.NativeAln(path, only)

    def pileup_flags(self, mq, region):
        """What window_adjust extracts from the pileup (StrainCall.cpp:702-736)."""
        name, a0, b0 = parse_region(region)
        return self.native.pileup_flags(mq, name, a0, b0)
