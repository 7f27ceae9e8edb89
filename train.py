#!/usr/bin/env python3
"""Entry point with the reference's CLI (train.py:7-32 of the reference):

    python train.py --config config/train_denet_synthetic.yml
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py --config ...
"""
import argparse

import yaml

from denet_amd.trainer import get_trainer


def arg_parser():
    parser = argparse.ArgumentParser(description='config')
    parser.add_argument('--config', type=str, default='config/train_denet_synthetic.yml',
                        help='Configuration file to use.')
    return parser.parse_args()


if __name__ == '__main__':
    arg = arg_parser()
    with open(arg.config) as config_file:
        config = yaml.load(config_file, Loader=yaml.SafeLoader)
    trainer = get_trainer(config['train']['trainer'])(config)
    trainer.train()
