"""Build-container-only stand-in for the third-party ``cyvcf2`` wheel (absent
from this image) so that the *reference* (/root/reference) can be imported to
generate golden vectors (tools/gen_golden.py).  It simply re-exports this
repo's own VCF decoder under cyvcf2's names.  Never shipped, never imported by
the product or by the tests."""
from trtools_amd.vcfio import VCFReader as VCF, Variant, VCFWriter as Writer  # noqa: F401
