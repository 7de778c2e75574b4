"""Empty stand-in so `import mchap` succeeds without pysam (golden generation only)."""


class AlignmentFile:  # pragma: no cover
    pass


class VariantFile:  # pragma: no cover
    pass


class FastaFile:  # pragma: no cover
    pass
