"""Stand-in for the absent `py-cpuinfo` wheel so that /root/reference can be imported
as a checker when generating golden vectors (test infrastructure only; never shipped).

The reference touches exactly two members (utils/device.py:417-437, inference/backend.py:39):
`get_cpu_info()` and `CPUID().get_max_extension_support()/_run_asm()`.
"""


def _flags():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return line.split(":", 1)[1].split()
    except OSError:
        pass
    return []


def get_cpu_info():
    import os

    return {"brand_raw": "generic x86_64", "flags": _flags(), "arch": "X86_64", "count": os.cpu_count() or 1}


class CPUID:
    def get_max_extension_support(self):
        return 7

    def _run_asm(self, *args):
        # bit 5 of CPUID.(EAX=7,ECX=1).EAX == AVX512_BF16
        return (1 << 5) if "avx512_bf16" in _flags() else 0
