"""trtools_amd -- MI355X-native implementation of the TRTools statSTR / dumpSTR
per-locus hot path.  ``__version__`` tracks the reference
release whose behaviour is reproduced."""
__version__ = "6.1.0+mi355x.1"
