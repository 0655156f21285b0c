"""CPU checks of the host-side generator of the full-file configuration: its counter-based PRNG
equals the device generator's and the oracle's, its Blocks have the widths the configuration names,
and an oracle-built image of them indexes as 4 Int arrays per Block."""
import numpy as np


class OracleCodec:
    @staticmethod
    def compress_int_array(v):
        from oracle import zbra_oracle_c as oc
        return oc.compress_int_array(v)


def test_host_prng_equals_device_prng_and_oracle():
    import torch
    from oracle import zbra_oracle_np as on
    from zbra_amd import workloads
    i = np.arange(123456, 123456 + 5000, dtype=np.uint64)
    for seed in (0x5EED0005, 0x5EED1005, 0x5EED0002):
        host = workloads.sm_np(seed, i)
        assert np.array_equal(host, on.sm(seed, i))
        dev = workloads.sm(seed, torch.from_numpy(i.astype(np.int64))).numpy().view(np.uint64)  # torch on the CPU: same code as on the GPU
        assert np.array_equal(host, dev)


def test_file_blocks_shape_and_widths():
    from oracle import zbra_oracle_c as oc
    from zbra_amd import binary as zb
    from zbra_amd import workloads
    rows = 1 << 12
    blocks = workloads.file_blocks(7, 2, rows)
    assert [b.row_count for b in blocks] == [rows, rows]
    names = [name for name, _ in blocks[0].table.column.fields]
    assert tuple(names) == workloads.FILE_COLUMNS == tuple(sorted(names))     # alphabetical, as the reference CLI emits struct keys
    cols = workloads.file_block_columns(7, rows)
    assert np.array_equal(dict(blocks[0].table.column.fields)["value"].values, cols["value"])
    assert np.all(np.diff(cols["time"]) == 1000) and cols["time"][0] == 1_700_000_000_000 + 7 * rows * 1000
    assert cols["attribute_id"].min() >= 0 and cols["attribute_id"].max() < 200
    widths = {k: oc.compress_int_array(v)[12] for k, v in cols.items()}
    assert widths["attribute_id"] == 8 and widths["value"] == 12 and widths["entity_id"] <= 8 and widths["time"] == 22
    # neighbouring Blocks continue the same table
    nxt = workloads.file_block_columns(8, rows)
    assert nxt["time"][0] == cols["time"][-1] + 1000
    img = zb.BinaryFile("{}", blocks).to_bytes(OracleCodec)
    info, arrays = zb.file_index(img)
    assert info.block_count == 2 and info.n_arrays == 8 and info.total_values == 8 * rows
    assert all(arrays[k].role == 0 and arrays[k].count == rows and arrays[k].block == k // 4 for k in range(8))
