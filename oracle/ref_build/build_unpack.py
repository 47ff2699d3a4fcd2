"""Builds oracle/_ref/unpack_vocab from our driver + the reference's own QuickLZ decoder, compiled where it lies
(/root/reference/third_party/DBow3/src/quicklz.c).  Test infrastructure only; needs /root/reference (this container)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/third_party/DBow3"
OUT = os.path.join(HERE, "..", "_ref")


def build():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present: the vocabulary fixture can only be regenerated in the build container")
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "unpack_vocab")
    cmd = ["gcc", "-O2", "-I", os.path.join(REF, "include"), "-I", os.path.join(REF, "include", "DBow3"), os.path.join(HERE, "unpack_vocab.c"),
           os.path.join(REF, "src", "quicklz.c"), "-o", exe]
    subprocess.check_call(cmd)
    return exe


if __name__ == "__main__":
    print(build())
