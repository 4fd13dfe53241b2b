"""Static discipline check of the hand-counted rings in tower_seq.hip, on the compiler's .s (cdna_hip_programming.md 5.7:
hipcc neither counts the memory operations of an `asm` statement nor keeps out of the registers they are still loading).

For every tower_seq_kernel instantiation, between the `; TW_STREAM_BEGIN` and `; TW_STREAM_END` markers:
  1. the only vector-memory instructions are the LDS-DMAs of the weight ring (a global load / store / scratch access in there
     would enter the vmcnt queue and make `s_waitcnt vmcnt(8)` wait for the wrong operations);
  2. between two chunk waits (`s_waitcnt vmcnt(N)` in an asm block) exactly 8 LDS-DMAs are issued (or none, at the tail);
  3. an asm `ds_read_b128` destination is not read or written by ANY instruction until a wait that covers it has been
     executed (covered: at least N LDS operations were issued after it when `s_waitcnt lgkmcnt(N)` runs - LDS operations
     complete in order), and no such read is pending at a branch or a branch target;
  4. the kernel has no scratch (a spill would be vector-memory traffic inside the stream: see 1).
Usage: check_asm_ring.py build/asm/tower_seq.s   (exit code 1 on a violation)"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
VMEM = re.compile(r"^(global_|buffer_|scratch_|flat_)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_ZN5dldkd2tw16tower_seq_kernel\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(ln)
            if ln.startswith(".Lfunc_end"):            # (not the first s_endpgm: a kernel may have early exits)
                yield name, body
                name = None


def check_kernel(name, body):
    errs = []
    inside = False
    in_asm = False
    pending = []          # [dest regs, index of the read among LDS ops, line number]
    n_lds = 0
    dma_since_wait = None
    n_reads = n_dma = 0
    for i, raw in enumerate(body):
        ln = raw.strip()
        if "TW_STREAM_BEGIN" in ln:
            inside, dma_since_wait = True, None
            continue
        if "TW_STREAM_END" in ln:
            if pending:
                errs.append(f"{name}: {len(pending)} asm reads pending at the end of the stream")
            inside = False
            continue
        if ln.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ln.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not inside or not ln or ln.startswith(";") or ln.startswith("."):
            if inside and re.match(r"^\.LBB\w+:", ln) and pending:
                errs.append(f"{name}:{i}: asm read pending at a branch target")
            continue
        op = ln.split()[0]
        if re.match(r"^s_cbranch|^s_branch", op) and pending:
            errs.append(f"{name}:{i}: asm read pending at a branch ({ln})")
        # 3. nobody touches a register that an asm read is still loading
        used = regs_of(ln.split(";")[0])
        is_ring_read = in_asm and op == "ds_read_b128"
        for dest, _, at in pending:
            if used & dest:
                errs.append(f"{name}:{i}: `{ln}` touches v{sorted(dest)[0]}.. loaded by the asm read at line {at} before its wait")
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ln)
            if m:
                nw = int(m.group(1))
                pending = [p for p in pending if n_lds - 1 - p[1] < nw]
            m = re.search(r"vmcnt\((\d+)\)", ln)
            if m and in_asm:
                if dma_since_wait is not None and dma_since_wait not in (0, 8):
                    errs.append(f"{name}:{i}: {dma_since_wait} LDS-DMAs between two chunk waits (expected 8)")
                dma_since_wait = 0
        if op.startswith("ds_"):
            if is_ring_read:
                dest = regs_of(ln.split(",")[0])
                pending.append((dest, n_lds, i))
                n_reads += 1
            n_lds += 1
        if VMEM.match(op):
            if op != "global_load_lds_dwordx4":
                errs.append(f"{name}:{i}: vector-memory instruction inside the weight stream: {ln}")
            else:
                n_dma += 1
                if dma_since_wait is not None:
                    dma_since_wait += 1
    return errs, n_reads, n_dma


def main(path):
    lines = open(path).read().splitlines()
    errs = []
    found = 0
    for name, body in kernels(lines):
        found += 1
        e, n_reads, n_dma = check_kernel(name, body)
        if n_reads == 0 or n_dma == 0:
            e.append(f"{name}: no ring reads / LDS-DMAs found between the stream markers (the audit would be vacuous)")
        errs += e
        print(f"{name}: {n_reads} ring reads, {n_dma} LDS-DMAs in the stream, {len(e)} violations")
    text = "\n".join(lines)
    for m in re.finditer(r"\.name:\s+(_ZN5dldkd2tw16tower_seq_kernel\w+)\n(?:.*\n){0,12}?\s+\.private_segment_fixed_size:\s+(\d+)", text):
        if int(m.group(2)) != 0:
            errs.append(f"{m.group(1)}: scratch {m.group(2)} bytes per lane")
    if not found:
        errs.append("no tower_seq_kernel found")
    for e in errs[:40]:
        print("VIOLATION", e)
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
