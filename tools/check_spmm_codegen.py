#!/usr/bin/env python3
"""Build-time guard for wagg_spmm.hip: between the inline-asm statements of the chunk loop the entry-list
registers (v[3:35], hard-coded in tools/gen_spmm_asm.py) hold loads that are still in flight, so the
compiler-generated code between two statements of one item must not touch v[3:35] (scalar code, or
vector code on other registers such as the accumulator zeroing) and the kernel must not use scratch.  Reads the device assembly of spmm_kernel (hipcc -S) and fails loudly otherwise.

usage: check_spmm_codegen.py <wagg_spmm.s>"""
import re
import sys

txt = open(sys.argv[1]).read()
V_LAST = 35          # tools/gen_spmm_asm.py: private registers v[3:V_LAST]
kernels = re.findall(r"^(_ZN4wagg11spmm_kernel\w*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M)
if len(kernels) != 2:
    sys.exit("expected spmm_kernel<float> and spmm_kernel<double> in %s, found %d" % (sys.argv[1], len(kernels)))
for kname, body in kernels:
  lines = body.splitlines()
  if any("scratch_" in l for l in lines):
      sys.exit(kname + ": spmm_kernel spills to scratch: the pinned accumulators / in-flight list registers are not safe")
  # statements in order; the protected span = from the end of SPMM_LOAD_LIST_ASM (first statement that issues
  # global_load_dword into the list registers) to the end of the last chunk statement of the loop body
  stmts, cur, start = [], None, None
  for i, l in enumerate(lines):
      if ";;#ASMSTART" in l:
          cur, start = [], i
      elif ";;#ASMEND" in l and cur is not None:
          stmts.append((start, i, cur))
          cur = None
      elif cur is not None:
          cur.append(l.strip())
  list_stmts = [k for k, (_, _, body) in enumerate(stmts) if any(b.startswith("global_load_dword v") for b in body)]
  if len(list_stmts) < 3:
      sys.exit("expected the list-load statement and two chunk statements, found %d" % len(list_stmts))
  first, last = list_stmts[0], list_stmts[-1]
  bad = []
  spans = [(stmts[k][1] + 1, stmts[k + 1][0]) for k in range(first, min(last + 1, len(stmts) - 1))]
  # ... and the loop latch behind the barrier that follows the last chunk statement (the next list sits in
  # set A there), up to the branch that closes the loop
  tail = stmts[min(last + 1, len(stmts) - 1)][1] + 1
  end = tail
  while end < len(lines) and not lines[end].strip().startswith(("s_cbranch", "s_branch")):
      end += 1
  spans.append((tail, end))
  for a, b in spans:
      for l in lines[a:b]:
          op = l.strip().split()[0] if l.strip() and not l.strip().startswith((";", ".")) else ""
          if op and not op.startswith(("s_", ".")) and not op.endswith(":"):
              regs = [int(x) for x in re.findall(r"\bv(\d+)\b", l)]
              for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", l):
                  regs += list(range(int(lo), int(hi) + 1))
              if any(3 <= r <= V_LAST for r in regs) or not regs:
                  bad.append(l.strip())
  if bad:
      sys.exit("vector code on v[3:35] between spmm_kernel's statements (list registers are live there):\n  " + "\n  ".join(bad[:20]))
  print(kname + ": %d statements checked, glue code leaves v[3:35] alone, no scratch" % (last - first + 1))
