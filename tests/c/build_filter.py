"""Build tests/c/_build/vf_harness: the reference's UNMODIFIED ffmpeg filter (Transform360/vf_transform360.c, compiled
where it lies under /root/reference against the test-only libav stand-ins in tests/c/avstub) + tests/c/vf_harness.c,
linked against libTransform360.so the way ffmpeg's --extra-libs='-lTransform360 -lstdc++' does.

The reference is not present on the GPU box: the filter OBJECT built here (tests/c/_build/, git-ignored) travels
with the snapshot, the reference source is never copied."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BUILD = os.path.join(HERE, "_build")
REF_FILTER = "/root/reference/Transform360/vf_transform360.c"
FILTER_OBJ = os.path.join(BUILD, "vf_transform360.o")
HARNESS = os.path.join(BUILD, "vf_harness")


def build(lib_path):
    """Returns the harness path, or None when neither the reference nor a prebuilt filter object is available."""
    os.makedirs(BUILD, exist_ok=True)
    common = ["gcc", "-std=gnu11", "-O1", "-w", "-I", os.path.join(HERE, "avstub"), "-I", os.path.join(ROOT, "include")]
    if os.path.exists(REF_FILTER):
        subprocess.check_call(common + ["-c", REF_FILTER, "-o", FILTER_OBJ])
    elif not os.path.exists(FILTER_OBJ):
        return None
    hobj = os.path.join(BUILD, "vf_harness.o")
    subprocess.check_call(common + ["-c", os.path.join(HERE, "vf_harness.c"), "-o", hobj])
    libdir = os.path.dirname(lib_path)
    subprocess.check_call(["gcc", "-o", HARNESS, hobj, FILTER_OBJ, "-L", libdir, "-lTransform360", "-lstdc++", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return HARNESS
