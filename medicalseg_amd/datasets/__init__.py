from .dataset import MedicalDataset, LungCoronavirus, MRISpineSeg, SyntheticCT, DataLoader
