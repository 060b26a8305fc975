"""`StrainCall` entry point for MI355X -- same command line and stdout as the
reference executable (/root/reference/StrainCall/StrainCall.cpp:34-55 help,
:98-154 argv, :972-1059 main), as launched by scripts/rambl.py:181-187:

    StrainCall -r <name>:<p0>-<p1> -q INT -D INT -I INT -l INT -t FLT -d FLT -w INT
               <gene.fasta> <reads.bam|reads.sam>

The graph and clustering stages run on the GPU through libstraincall_hip.so;
there is no CPU fallback.  Environment (not part of the reference's argv, which
treats unknown tokens as file names): SC_DEVICE=<hip ordinal>,
SC_TRACE_FILE=<path> (per-level strain/abundance trace), SC_STATS=1 (timing on
stderr).
"""
import os
import sys

import numpy as np

from . import ingest, samio

HELP = """StrainCall marker_gene read_mapping
           [-r gn:p0-p1] [-w window_size]
           [-e error_rate] [-q map_qual]

Options
-r,--roi           region of interesting, gn is gene name,
                   p0 is starting position, p1 is ending position (inclusive)
-w,--window        the size of scanning window [500]
-o,--overlap       the size of window-window overlap [100]
-e,--error-rate    sequencing error rate [0.01]
-D,--max-depth     downsample data to the specified depth [800]
-q,--map-qual      only include reads with mapping quality >= INT [3]
-I,--max-ins       only include reads with insertions <= INT [10]
-l,--read-len      only include reads with length >=INT [80]
-t,--tau           only include strains with abundance level >=FLT [0.02]
-d,--diff-rate     only include strains with difference rate >=FLT [0.01]
-G,--plot-graph    print graph
-h,--help          print this message

"""


class Parameters:
    """sc_parameter, StrainCall.cpp:58-95."""

    def __init__(self):
        self.gene_file = ""
        self.mapping_file = ""
        self.roi = ""
        self.window_size = 500
        self.overlap_size = 100
        self.error_rate = np.float32(0.01)
        self.mapping_qual = 3
        self.max_ins = 10
        self.read_len = 80
        self.print_help = False
        self.d0 = 0
        self.d1 = 0
        self.tau = np.float32(0.02)
        self.diff_rate = np.float32(0.01)
        self.max_depth = 800
        self.plot_graph = False


def _stof(s):
    import re
    m = re.match(r"\s*[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?|inf|nan)", s, re.I)
    if not m:
        raise ValueError("stof: no conversion for %r" % (s,))
    return np.float32(float(m.group(0)))


def parse_cmd_line(argv):
    """sc_parse_cmd_line, StrainCall.cpp:98-154: options as -x, --long or -long;
    first bare token = gene FASTA, every later bare token = mapping file."""
    pa = Parameters()
    p = 0
    i = 0
    n = len(argv)

    def is_(op, short, long_):
        return op == short or op == "--" + long_ or op == "-" + long_

    while i < n:
        op = argv[i]
        if op[:1] == "-":
            if is_(op, "-h", "help"):
                pa.print_help = True
            elif is_(op, "-r", "roi"):
                i += 1; pa.roi = argv[i]
            elif is_(op, "-w", "window"):
                i += 1; pa.window_size = ingest.stoi(argv[i])
            elif is_(op, "-e", "error-rate"):
                i += 1; pa.error_rate = _stof(argv[i])
            elif is_(op, "-q", "map-qual"):
                i += 1; pa.mapping_qual = ingest.stoi(argv[i])
            elif is_(op, "-o", "overlap"):
                i += 1; pa.overlap_size = ingest.stoi(argv[i])
            elif is_(op, "-l", "read-len"):
                i += 1; pa.read_len = ingest.stoi(argv[i])
            elif is_(op, "-t", "tau"):
                i += 1; pa.tau = _stof(argv[i])
            elif is_(op, "-d", "diff-rate"):
                i += 1; pa.diff_rate = _stof(argv[i])
            elif is_(op, "-D", "max-depth"):
                i += 1; pa.max_depth = ingest.stoi(argv[i])
            elif is_(op, "-I", "max-ins"):
                i += 1; pa.max_ins = ingest.stoi(argv[i])
            elif is_(op, "-G", "plot-graph"):
                pa.plot_graph = True
        else:
            if p == 0:
                pa.gene_file = op
            else:
                pa.mapping_file = op
            p += 1
        i += 1
    return pa


def load_regions(pa):
    """Windows + per-window ingest: StrainCall.cpp:988-1012."""
    fasta = samio.Fasta(pa.gene_file)
    fai = samio.read_fai(pa.gene_file + ".fai")
    aln = samio.Alignments(pa.mapping_file)
    windows = ingest.make_scan_window(pa, fai, aln)
    out = []
    for gn, p0, p1 in windows:
        roi = "%s:%d-%d" % (gn, p0, p1)
        gene_seq = fasta.fetch(roi)
        reads = ingest.load_mapping_reads(gene_seq, aln, pa.mapping_qual, pa.read_len, pa.max_ins, pa.max_depth, roi)
        out.append(((gn, p0, p1), reads))
    return out


def format_fasta(window, result, tau):
    """StrainCall.cpp:1033-1046."""
    gn, p0, p1 = window
    out = []
    t = float(np.float32(tau))
    for si, (seq, ab) in enumerate(zip(result.seqs, result.abundance)):
        if ab >= t:
            out.append(">contig%s%d%d%d\n%s\n" % (gn, p0, p1, si, seq))
    return "".join(out)


def main(argv=None, out=None, err=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    out = out or sys.stdout
    err = err or sys.stderr
    pa = parse_cmd_line(argv)
    if pa.print_help or len(argv) == 0:
        err.write(HELP)
        return 0
    from . import capi            # fails loudly when the HIP library is missing
    regions = load_regions(pa)
    dev = int(os.environ.get("SC_DEVICE", "0"))
    trace_file = os.environ.get("SC_TRACE_FILE")
    params = capi.default_params(float(pa.error_rate), float(pa.tau), float(pa.diff_rate),
                                 graph_only=pa.plot_graph, want_trace=bool(trace_file))
    traces = []
    with capi.Context(dev, 1) as ctx:
        for window, reads in regions:
            if len(reads) == 0:     # StrainCall.cpp:1009
                continue
            res = ctx.run(reads, params, want_graph=pa.plot_graph, want_trace=bool(trace_file))
            if pa.plot_graph:
                out.write(res.graph)
            else:
                out.write(format_fasta(window, res, pa.tau))
            if trace_file and res.trace is not None:
                traces.append(res.trace)
            if os.environ.get("SC_STATS"):
                err.write("sc_stats %s %s\n" % ("%s:%d-%d" % window, res.stats))
    if trace_file:
        with open(trace_file, "w") as f:
            f.write("".join(traces))
    out.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
