"""Importable alias for the product package.

The product lives in ``restrictive-hierarchical-semantic-segmentation_amd/`` (a
directory name that is not a valid Python identifier), so this three-line
package points its ``__path__`` there: ``import hrseg_amd.Models.models`` loads
``restrictive-hierarchical-semantic-segmentation_amd/Models/models.py``.
"""
import os as _os

PACKAGE_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                            "restrictive-hierarchical-semantic-segmentation_amd")
__path__ = [PACKAGE_DIR]
