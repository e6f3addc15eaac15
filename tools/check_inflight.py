#!/usr/bin/env python3
"""Check hand-scheduled loads in a kernel's ISA: no instruction may touch the destination
registers of an inline-asm global load between the load and the s_waitcnt that covers it.

conv_stack.hip's loader waves issue their row loads from inline asm and wait with explicit
`s_waitcnt vmcnt(N)` (hipcc would wait for vmcnt(0), see the comment there).  hipcc does not know
that those registers are still being written, so nothing stops it from copying or reusing them
early; this script replays the instruction stream (vmcnt retires in order) and reports any such
use, and any label or branch reached with such a load in flight.

usage: tools/check_inflight.py [file.s | file.hip]   (a .hip file is compiled with hipcc -S)
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
VMEM = re.compile(r'^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|flat_load|flat_store)')
LOAD = re.compile(r'^\s*global_load_dword(x[234])?\s+(v\[(\d+):(\d+)\]|v(\d+)),')
WAIT = re.compile(r'vmcnt\((\d+)\)')
REGS = re.compile(r'v\[(\d+):(\d+)\]|\bv(\d+)\b')


def registers(text):
    found = set()
    for low, high, single in REGS.findall(text):
        if single:
            found.add(int(single))
        else:
            found.update(range(int(low), int(high) + 1))
    return found


def assembly(path):
    path = Path(path)
    if path.suffix == '.s':
        return path.read_text()
    with tempfile.TemporaryDirectory() as scratch:
        out = Path(scratch) / 'kernel.s'
        subprocess.run(
            ["/opt/rocm/bin/hipcc", "-O3", '--offload-arch=gfx950', '-std=c++17', f'-I{ROOT}/include',
             f'-I{ROOT}/emphases_amd/csrc', '-S', '--cuda-device-only', str(path), '-o', str(out)],
            check=True, stderr=subprocess.DEVNULL)
        return out.read_text()


def check(text):
    """-> (asm loads seen, list of violations)"""
    problems, seen = [], 0
    issued = 0                # vector-memory operations issued so far, in order
    flying = []               # (issue number, registers, line number)
    in_asm = False
    function = None
    for number, line in enumerate(text.splitlines(), 1):
        code = line.split(';')[0].rstrip() if not line.lstrip().startswith(';;#') else line
        if ';;#ASMSTART' in line:
            in_asm = True
            continue
        if ';;#ASMEND' in line:
            in_asm = False
            continue
        if not code.strip():
            continue
        if re.match(r'^[A-Za-z_.$][\w.$]*:', code):
            if not code.startswith('.'):          # a new function
                function, issued, flying = code.rstrip(':'), 0, []
            elif flying:
                problems.append(f'{function}: line {number}: label {code.strip()} with '
                                f'{len(flying)} asm load(s) in flight')
                flying = []
            continue
        for wait in WAIT.findall(code):
            flying = [entry for entry in flying if entry[0] > issued - int(wait)]
        touched = registers(code)
        load = LOAD.match(code) if in_asm else None
        for _, busy, origin in flying:
            if touched & busy:
                problems.append(f'{function}: line {number}: `{code.strip()}` touches '
                                f'v{sorted(touched & busy)} of the asm load at line {origin}')
        if re.match(r'^\s*(s_cbranch|s_branch|s_setpc|s_swappc)', code) and flying:
            problems.append(f'{function}: line {number}: `{code.strip()}` with '
                            f'{len(flying)} asm load(s) in flight')
        if VMEM.match(code):
            issued += 1
            if load:
                seen += 1
                if load.group(5):
                    target = {int(load.group(5))}
                else:
                    target = set(range(int(load.group(3)), int(load.group(4)) + 1))
                flying.append((issued, target, number))
    return seen, problems


def main():
    source = sys.argv[1] if len(sys.argv) > 1 else ROOT / 'emphases_amd/csrc/conv_stack.hip'
    seen, problems = check(assembly(source))
    for problem in problems:
        print(problem)
    print(f'{seen} inline-asm loads checked, {len(problems)} problem(s)')
    return 1 if problems or not seen else 0


if __name__ == '__main__':
    sys.exit(main())
