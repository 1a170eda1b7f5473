#!/usr/bin/env python3
"""Write the two headers the reference sources expect from their configure step
(`config.h`, `libavutil/avconfig.h`) WITHOUT running the reference build system.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Output goes to oracle/_ref/cfg/
(git-ignored).  The reference tree is only *read*: every `ARCH_*`, `HAVE_*`,
`CONFIG_*` token that appears in the directories we compile from is defined to
0, then a short allow-list (plain C on x86-64 Linux/glibc, no asm, no threads,
no external libraries) is set to 1.  Nothing is copied out of the reference.
"""
import os
import re
import sys

REF = os.environ.get("LIBAV_REF", "/root/reference")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "_ref", "cfg")

SCAN_DIRS = ["libavcodec", "libavutil", "libswscale", "libavformat"]

# What a "--disable-asm --disable-pthreads" configure run would find on this image.
ONES = """
ARCH_GENERIC_NOT
HAVE_FAST_64BIT HAVE_FAST_UNALIGNED HAVE_FAST_CLZ HAVE_FAST_CMOV HAVE_LOCAL_ALIGNED
HAVE_ATTRIBUTE_PACKED HAVE_ATTRIBUTE_MAY_ALIAS HAVE_PRAGMA_DEPRECATED HAVE_INLINE_ASM_LABELS
HAVE_BUILTIN_VECTOR
HAVE_ALIGNED_STACK HAVE_POSIX_MEMALIGN HAVE_MEMALIGN
HAVE_CBRT HAVE_CBRTF HAVE_COPYSIGN HAVE_ERF HAVE_EXP2 HAVE_EXP2F HAVE_EXPF HAVE_HYPOT
HAVE_ISFINITE HAVE_ISINF HAVE_ISNAN HAVE_LDEXPF HAVE_LLRINT HAVE_LLRINTF HAVE_LOG10F HAVE_LOG2
HAVE_LOG2F HAVE_LRINT HAVE_LRINTF HAVE_POWF HAVE_RINT HAVE_ROUND HAVE_ROUNDF HAVE_SINF HAVE_TRUNC
HAVE_TRUNCF HAVE_ATANF HAVE_ATAN2F HAVE_COSF
HAVE_UNISTD_H HAVE_SYS_TIME_H HAVE_SYS_RESOURCE_H HAVE_SYS_SELECT_H HAVE_SYS_PARAM_H HAVE_SYS_MMAN_H
HAVE_IO_H_NOT HAVE_FCNTL HAVE_GETTIMEOFDAY HAVE_CLOCK_GETTIME HAVE_GMTIME_R HAVE_LOCALTIME_R
HAVE_ISATTY HAVE_NANOSLEEP HAVE_USLEEP HAVE_SCHED_GETAFFINITY HAVE_SYSCONF HAVE_STRERROR_R
HAVE_MKSTEMP HAVE_MMAP HAVE_MPROTECT HAVE_LSTAT HAVE_GETOPT HAVE_SETRLIMIT
HAVE_STRUCT_STAT_ST_MTIM_TV_NSEC HAVE_DIRENT_H HAVE_POLL_H HAVE_ARPA_INET_H
HAVE_SYMVER HAVE_SYMVER_ASM_LABEL HAVE_RDTSC_NOT
HAVE_ATOMICS_GCC HAVE_STDATOMIC_H_NOT
CONFIG_AVUTIL CONFIG_AVCODEC CONFIG_AVFORMAT CONFIG_SWSCALE
CONFIG_SAFE_BITSTREAM_READER CONFIG_FAST_UNALIGNED CONFIG_ERROR_RESILIENCE_NOT
CONFIG_H264_DECODER CONFIG_H264_PARSER CONFIG_HEVC_DECODER CONFIG_HEVC_PARSER
CONFIG_H264DSP CONFIG_H264CHROMA CONFIG_H264PRED CONFIG_H264QPEL CONFIG_H264PARSE CONFIG_VIDEODSP
CONFIG_CABAC CONFIG_GOLOMB CONFIG_BSWAPDSP CONFIG_STARTCODE
CONFIG_H264_MP4TOANNEXB_BSF CONFIG_NULL_BSF
CONFIG_SWSCALE_ALPHA
""".split()
ONES = {t for t in ONES if not t.endswith("_NOT")}

PUB = ["BIGENDIAN", "FAST_UNALIGNED"]  # AV_HAVE_* in avconfig.h


def main():
    toks = set()
    pat = re.compile(r"\b((?:ARCH|HAVE|CONFIG)_[A-Z0-9_a-z]+)\b")
    for d in SCAN_DIRS:
        root = os.path.join(REF, d)
        for dirpath, _dirs, files in os.walk(root):
            for f in files:
                if not f.endswith((".c", ".h")):
                    continue
                try:
                    with open(os.path.join(dirpath, f), errors="replace") as fh:
                        toks.update(pat.findall(fh.read()))
                except OSError:
                    pass
    toks = {t for t in toks if re.fullmatch(r"[A-Z0-9_]+", t)}
    toks |= ONES
    os.makedirs(os.path.join(OUT, "libavutil"), exist_ok=True)
    with open(os.path.join(OUT, "config.h"), "w") as o:
        o.write("/* written by oracle/mk_refcfg.py (NOT by the reference's configure) */\n")
        o.write("#ifndef LIBAV_CONFIG_H\n#define LIBAV_CONFIG_H\n")
        o.write('#define LIBAV_CONFIGURATION "oracle: plain C, no asm, no threads"\n')
        o.write('#define LIBAV_LICENSE "LGPL version 2.1 or later"\n')
        o.write('#define AVCONV_DATADIR "/nonexistent"\n')
        o.write('#define CC_IDENT "gcc"\n#define EXTERN_PREFIX ""\n#define EXTERN_ASM\n#define SLIBSUF ".so"\n')
        for t in sorted(toks):
            o.write("#define %s %d\n" % (t, 1 if t in ONES else 0))
        o.write("#endif\n")
    with open(os.path.join(OUT, "libavutil", "avconfig.h"), "w") as o:
        o.write("#ifndef AVUTIL_AVCONFIG_H\n#define AVUTIL_AVCONFIG_H\n")
        for p in PUB:
            o.write("#define AV_HAVE_%s %d\n" % (p, 1 if ("HAVE_" + p) in ONES else 0))
        o.write("#endif\n")
    with open(os.path.join(OUT, "avversion.h"), "w") as o:      # what avbuild/version.sh would write
        o.write('#define LIBAV_VERSION "oracle-build"\n')
    print("wrote", OUT, len(toks), "macros")


if __name__ == "__main__":
    main()
