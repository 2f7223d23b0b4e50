"""Diagnostic builds of the library (tools/ only): the product's own Makefile with extra flags, into a file of the caller's choice."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reinforcement_learning_rendezvous_amd", "csrc")


def build_variant(out, extra=()):
    """`make` both translation units with `extra` compiler flags (e.g. ["-DRDV_STAMPS"]) and link them into `out`."""
    obj = tempfile.mkdtemp(prefix="rdv_obj_")
    subprocess.check_call(["make", "-C", CSRC, "-B", "-j3", f"OUT={os.path.abspath(out)}", f"OBJ={obj}", "EXTRA=" + " ".join(extra)],
                          stdout=subprocess.DEVNULL)
    return out
