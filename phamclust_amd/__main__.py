"""``python -m phamclust_amd <infile> <outdir> [flags]`` runs the pipeline."""
import sys

from phamclust_amd.scripts import phamclust as pipeline

sys.exit(pipeline.main())
