"""phamclust_amd: MI355X-native pairwise genome-similarity matrix fill behind phamclust's
``METRICS`` / ``matrix_de_novo`` boundary.  See DESIGN.md."""

DATE = "2024-12-18"          # reference snapshot this build tracks (phamclust 1.3.3)
__version__ = "0.1.0"
