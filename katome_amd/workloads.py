"""The synthetic workloads of BASELINE.json / SURVEY.md 8(d) (definition: DESIGN.md 'Synthetic workload')."""
from dataclasses import dataclass


@dataclass(frozen=True)
class Workload:
    name: str
    reads: int
    read_len: int
    k: int
    genome_len: int
    err_rate: float
    n_inject_percent: int
    reverse_complement: bool = True

    @property
    def windows_per_read(self):
        return self.read_len - self.k + 1

    @property
    def stride(self):
        return (self.read_len + 3) // 4

    def expected_distinct_canonical(self):
        """rough count of distinct k-mers (one per strand pair): genome + error k-mers; sizes the table"""
        err = self.reads * self.read_len * self.err_rate * min(self.k, self.windows_per_read)
        return int(min(self.reads * self.windows_per_read, self.genome_len + err))

    def scaled(self, reads):
        """same coverage and error model on fewer reads (genome scaled with the read count)"""
        g = max(self.read_len * 4, int(self.genome_len * (reads / self.reads)))
        return Workload("%s/%d" % (self.name, reads), reads, self.read_len, self.k, g, self.err_rate,
                        self.n_inject_percent, self.reverse_complement)


WORKLOADS = {
    # BASELINE.json configs[1]: 1M synthetic 150 bp reads, k=31, bit-exact edge set vs CPU
    "c2": Workload("c2", 1_000_000, 150, 31, 1_000_000, 1e-3, 1),
    # configs[2]/[3]: 200M synthetic 150 bp reads, k=31 (the configuration the metric is quoted on)
    "c3": Workload("c3", 200_000_000, 150, 31, 100_000_000, 1e-3, 0),
    # configs[4]: 1B reads, k=63 (128-bit keys), 8 GPUs
    "c5": Workload("c5", 1_000_000_000, 150, 63, 1_000_000_000, 5e-4, 0),
    # not a BASELINE configuration: 101-bp reads, whose 71 windows are not a whole number of useful tiles
    # (counted as 5 tiles of 14 windows + 1 window, katome_tile_plan)
    # not a BASELINE configuration either: c3's reads at the k-mer size of the reference's example configuration
    # (config.txt: k_mer_size = 40) -- two-word k-mers, 111 windows = 4 tiles of 27 + 3 windows (katome_tile_plan)
    "k40": Workload("k40", 200_000_000, 150, 40, 100_000_000, 1e-3, 0),
    "r101": Workload("r101", 100_000_000, 101, 31, 50_000_000, 1e-3, 0),
}
