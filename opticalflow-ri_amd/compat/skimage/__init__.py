"""Stand-in for the one scikit-image name the reference's scripts import (``from skimage.io import imread``,
examples/PyHSchunck_Fs3_4_PyrLvls2.py:26), for machines without scikit-image.  Put this directory's parent at the END of
sys.path, so that an installed scikit-image is found first:

    sys.path.append(os.path.join(<package directory>, 'compat'))
"""
