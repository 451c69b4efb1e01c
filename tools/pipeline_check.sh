#!/bin/bash
# End-to-end run of the pieces around the device path on a synthetic pangenome (tools/simulate_pangenome.py):
#   PanGenie-index (host/index_builder) -> graph-only k-mer counts of a 30x sample -> counts into the index ->
#   HMM on the device (65 paths) -> genotyped VCF -> concordance with the sample's true genotypes.
# usage: tools/pipeline_check.sh <out dir> [length records samples coverage panel-seed sample-seed]
# (more than 50 samples = more than 100 paths: 15 haplotypes are sampled on the device first, as the reference does by default)
# The md5 of the VCF without its date line can be compared with the CPU twin's (tools/pipeline_cpu_check.py: oracle HMM).
set -e
OUT=${1:-gpurun_out/pipeline}; LEN=${2:-20000000}; REC=${3:-40000}; SAMPLES=${4:-32}; COV=${5:-30}; PSEED=${6:-11}; SSEED=${7:-5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/pg_pipeline.XXXXXX)
mkdir -p "$OUT"
{
  echo "pipeline_check: $LEN bases, $REC records, $SAMPLES panel samples (+ reference path), ${COV}x reads; $(nproc) host cores"
  t0=$(date +%s.%N)
  python "$ROOT/tools/simulate_pangenome.py" panel $LEN $REC $SAMPLES $PSEED $W/q
  python "$ROOT/tools/simulate_pangenome.py" sample $W/q $COV $SSEED
  t1=$(date +%s.%N); echo "simulated panel + reads: $(awk "BEGIN{printf \"%.1f\", $t1-$t0}") s ($(du -m $W/q_reads.fa | cut -f1) MB of reads)"
  PG_INDEX_VERBOSE=1 "$ROOT/tests/cpp/test_host.bin" index $W/q.fa $W/q.vcf $W/idx 31 0
  t2=$(date +%s.%N); echo "index built: $(awk "BEGIN{printf \"%.1f\", $t2-$t1}") s"
  "$ROOT/tests/cpp/test_host.bin" genotype $W/idx $W/q_reads.fa $W/out.vcf $(nproc)
  t3=$(date +%s.%N); echo "genotyped: $(awk "BEGIN{printf \"%.1f\", $t3-$t2}") s"
  python "$ROOT/tools/simulate_pangenome.py" score $W/q_truth.tsv $W/out.vcf
  echo "VCF without the date line: $(grep -v '^##fileDate' $W/out.vcf | md5sum | cut -d' ' -f1) ($(grep -vc '^#' $W/out.vcf) records)"
} 2>&1 | tee "$OUT/pipeline_check_${SAMPLES}samples.txt"
head -c 3000000 $W/out.vcf | gzip -c > "$OUT/pipeline_head_${SAMPLES}samples.vcf.gz"
rm -rf "$W"
