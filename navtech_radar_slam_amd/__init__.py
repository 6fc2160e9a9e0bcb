"""Import alias for the product package.

The product lives in the directory `navtech-radar-slam_amd/` (the name the project layout
prescribes); a hyphen cannot appear in a Python import statement, so this stub package simply
points its module search path there.  `import navtech_radar_slam_amd.synth` etc. resolve to
files in `navtech-radar-slam_amd/`.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "navtech-radar-slam_amd")
__path__.append(_PKG_DIR)
PACKAGE_DIR = _PKG_DIR
