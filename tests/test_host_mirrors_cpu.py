"""The host mirrors of haphic_amd/cluster.py (dict building in the reference's insertion order, the bin -> contig vote,
the inflation sweep with its files and log line, the .pairs front end, the blocked sweep) run on CPU against the SAME
frozen reference outputs as the GPU tests, with tests/oracle_lib.py standing in for the HIP library.  What this pins is
the Python around the kernels; the kernels themselves are pinned by the -m gpu tests."""
import pytest

from tests import oracle_lib
from tests import test_gpu_kernels as tk
from tests import test_gpu_pipeline as tp


@pytest.fixture
def host_only(monkeypatch):
    from haphic_amd import cluster
    monkeypatch.setattr(cluster, '_lib', oracle_lib)
    return cluster


@pytest.mark.parametrize('block_rows', [None, 37])
def test_toy_pipeline_files(host_only, golden_pipeline, tmp_path, block_rows):
    tp.test_cluster_files_byte_identical(golden_pipeline, tmp_path, block_rows)


def test_split_contigs_pipeline_files(host_only, tmp_path):
    tp.test_cluster_files_with_split_contigs(tmp_path)


def test_c1_pipeline_files(host_only, tmp_path):
    tp.test_cluster_files_c1_config(tmp_path)


def test_c4_allele_aware_pipeline_files(host_only, tmp_path):
    tp.test_cluster_files_c4_allele_aware(tmp_path)


def test_seven_containers(host_only, monkeypatch):
    tp.test_parse_alignments_seven_containers_allelic(monkeypatch)


def test_pairs_file_front_end(host_only, tmp_path, monkeypatch):
    tk.test_pairs_text_through_ingest(tmp_path, monkeypatch)


def test_wide_positions_mirror(host_only, tmp_path):
    tk.test_wide_positions_file_front_end_and_mirror(tmp_path)


@pytest.fixture
def host_only_everywhere(monkeypatch, host_only):
    import haphic_amd
    monkeypatch.setattr(haphic_amd, '_lib', oracle_lib)        # `from haphic_amd import _lib` inside the test bodies
    return host_only


def test_stat_fragments_mirror(host_only_everywhere):
    tp.test_re_sites_and_stat_fragments()


def test_filter_fragments_mirror(host_only_everywhere):
    tp.test_filter_fragments_and_rank_sums()


def test_link_weights_mirrors(host_only_everywhere):
    tp.test_link_weights_a6()


def test_reassign_group_link_sums_mirror(host_only_everywhere):
    tp.test_reassign_group_link_sums_f3()
