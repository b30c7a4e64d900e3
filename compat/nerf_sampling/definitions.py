"""ROOT_DIR: where dataset/, pretrained/, logs/ and experiments/configs/ live.  The reference anchors it at its own
package directory; here it is $NERF_SAMPLING_ROOT when set (point it at the reference checkout's nerf_sampling/
directory to reuse its yaml files and data layout), else this alias package's directory."""
import os

ROOT_DIR = os.environ.get("NERF_SAMPLING_ROOT") or os.path.dirname(os.path.abspath(__file__))
