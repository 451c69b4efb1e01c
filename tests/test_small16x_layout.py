"""CPU guard of the record / LDS-slot arithmetic of k_sweep_small16x (pangenie_amd/csrc/pg_small16x.h): the constants are read from
the header, the index tricks the kernel and k_records rely on are restated here — a silent edit of one side shows up without a GPU."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HDR = (ROOT / "pangenie_amd" / "csrc" / "pg_small16x.h").read_text()
DEV = (ROOT / "pangenie_amd" / "csrc" / "pg_device.h").read_text()
KER = (ROOT / "pangenie_amd" / "csrc" / "pg_kernels.hip").read_text()


def define(text, name):
    m = re.search(r"#define\s+%s\s+\(?(\d+)u?\)?" % name, text)
    assert m, name
    return int(m.group(1))


AMAX, XREC, XHDR, XCONST, XRO = define(DEV, "PG_AMAX"), define(HDR, "PG_XREC_BYTES"), define(HDR, "PG_XREC_HDR"), define(HDR, "PG_XREC_CONSTS"), define(HDR, "PG_XREC_ROWOFF")
STAB, SBYTES, BLMAX = define(HDR, "PG_XSLOT_TABLE"), define(HDR, "PG_XSLOT_BYTES"), define(HDR, "PG_XBLOCK_MAX")
ESTRIDE = AMAX + 1


def tri_local(a, b):   # pg_kernels.hip: tri_local
    return a * AMAX - a * (a - 1) // 2 + (b - a)


def x_pair_of(p):      # pg_small16x.h: x_pair_of
    a = (p >= 5) + (p >= 9) + (p >= 12) + (p >= 14)
    return a, p - (a * AMAX - a * (a - 1) // 2) + a


def test_record_layout_is_twelve_pieces():
    n_entries = AMAX * (AMAX + 1) // 2
    assert AMAX == 5 and n_entries == 15
    assert n_entries * 8 <= XHDR and XHDR % 16 == 0                      # the table entries, then 16-byte pieces
    assert (XHDR + 16, XCONST + 32, XRO + 16) == (XCONST, XRO, XREC)      # header, two pieces of constants, row offsets
    assert XREC == 12 * 16 and "p / 12u" in KER and "* 192u" in KER       # k_records writes 12 pieces per column


def test_pair_enumeration_is_tri_local_order():
    seen = []
    for p in range(15):
        a, b = x_pair_of(p)
        assert 0 <= a <= b < AMAX and tri_local(a, b) == p
        seen.append((a, b))
    assert len(set(seen)) == 15


def test_table_slot_and_row_offsets():
    row = ESTRIDE * 8                                    # 48 bytes per table row
    assert STAB == row and SBYTES == STAB + ESTRIDE * row   # a row of zeros in front, then the 6 x 6 table
    for a in range(AMAX):
        ro = (a + 1) * row                               # the row-offset byte of allele a
        assert ro < 256 and ((ro * 171) >> 10) - 8 == 8 * a   # the lane's table column out of its own row offset (read_xcol)
    # an entry parked at [a][b] and [b][a] never touches column 5 or the zero row: they stay zero for phantom alleles
    for p in range(15):
        a, b = x_pair_of(p)
        for r, c in ((a, b), (b, a)):
            off = STAB + r * row + c * 8
            assert STAB <= off < SBYTES and (off - STAB) % row != AMAX * 8


def test_block_geometry():
    for bl in (8, 6):                                    # store-only phases / phase 2
        pieces = bl * 12
        nq = (pieces + 15) // 16
        assert bl <= BLMAX and nq * 16 >= pieces and (bl % 3 == 0 or bl == 8)
        # the staging position of rel r in a backward block mirrors the forward one
        for r in range(3 * bl):
            fwd, bwd = r % bl, bl - 1 - r % bl
            assert 0 <= bwd < bl and fwd + bwd == bl - 1
        # step n needs rel n + 2: a block is parked one step before its first record is read, fetched a block earlier
        for n in range(4 * bl):
            p = (n + 2) % bl
            if p == bl - 1:
                assert (n + 3) % bl == 0                 # the next step is the first to read the parked block
