"""CPU: rambl.py stage-5 mirror -- region list, argv, LPT sharding, length filter,
and the N>1 gather of FASTA bytes with world_size 2 over gloo."""
import os
import subprocess
import sys
import textwrap

from rambl_amd import stage5

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roi_list_and_argv(tmp_path):
    fai = tmp_path / "seed_otus.fasta.fai"
    fai.write_text("otuA\t1502\t6\t60\t61\notuB\t1430\t1540\t60\t61\n")
    assert stage5.roi_list(str(fai)) == ["otuA:1-1502", "otuB:1-1430"]
    argv = stage5.straincall_argv("otuA:1-1502", "f.fa", "r.bam")
    assert argv == ["-r", "otuA:1-1502", "-q", "0", "-D", "800", "-I", "13", "-l", "70", "-t", "0.02", "-d", "0.02",
                    "-w", "5000", "f.fa", "r.bam"]


def test_lpt_and_length_filter():
    shards = stage5.lpt_shards([5, 9, 1, 7, 3, 3], 2)
    assert sorted(sum(shards, [])) == list(range(6))
    loads = [sum([5, 9, 1, 7, 3, 3][i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 2
    fa = ">a\n" + "A" * 399 + "\n>b\n" + "C" * 400 + "\n"
    assert stage5.seqtk_L(fa) == ">b\n" + "C" * 400 + "\n"


def test_gather_single_process():
    assert stage5.gather_fasta([">x\nAC\n", ">y\nGT\n"], [1, 0], 2) == ">y\nGT\n>x\nAC\n"


def test_gather_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent('''
        import os, sys
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from rambl_amd import stage5
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        r = dist.get_rank()
        units = [">u%%d\\n%%s\\n" %% (i, "ACGT" * (i + 1)) for i in range(5)]
        mine = stage5.lpt_shards([5.0, 4.0, 3.0, 2.0, 1.0], 2)[r]
        full = stage5.gather_fasta([units[i] for i in mine], mine, 5, dist)
        if r == 0:
            assert full == "".join(units), full
            open(%r, "w").write("ok")
        dist.barrier()
        dist.destroy_process_group()
    ''' % (ROOT, str(tmp_path / "ok"))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    assert (tmp_path / "ok").read_text() == "ok"
