#!/usr/bin/env python3
"""Entry point with the reference's CLI (test.py:7-32 of the reference):

    python test.py --config config/test_denet_synthetic.yml
"""
import argparse

import yaml

from denet_amd.trainer import get_trainer


def arg_parser():
    parser = argparse.ArgumentParser(description='config')
    parser.add_argument('--config', type=str, default='config/test_denet_synthetic.yml',
                        help='Configuration file to use.')
    return parser.parse_args()


if __name__ == '__main__':
    arg = arg_parser()
    with open(arg.config) as config_file:
        config = yaml.load(config_file, Loader=yaml.SafeLoader)
    trainer = get_trainer(config['test']['trainer'])(train_config=None, test_config=config)
    trainer.test()
