#!/usr/bin/env python
"""Inference driver with the reference's command line (reference test.py:9-37): one image pair at a time, no
shuffle, no flip, weights from `<checkpoints_dir>/<name>/<which_epoch>_net_{G,E}.pth`, every visual of
`model.get_current_visuals()` written as `<results_dir>/<name>/<phase>_<which_epoch>/images/<stem>_<label>.png`
(naming of the reference's `util/visualizer.py:13-35`).  The reference builds its result page with `dominate`, which
does not exist here: `index.html` is written as plain text with the same rows.

    python test.py --dataroot synthetic --model wsgan_emb --which_model_netG resnet_9blocks --name run --how_many 8
"""
import ntpath
import os

import numpy as np

from pcgan_amd.data import CreateDataLoader
from pcgan_amd.models import create_model
from pcgan_amd.options.test_options import TestOptions
from pcgan_amd.util import util


class ResultPage(object):
    """image directory + a table row per processed input (stands in for the reference's util/html.py)"""

    def __init__(self, web_dir, title):
        self.web_dir, self.title = web_dir, title
        self.img_dir = os.path.join(web_dir, 'images')
        util.mkdirs([self.web_dir, self.img_dir])
        self.rows = []

    def get_image_dir(self):
        return self.img_dir

    def add_row(self, header, names, labels, width):
        cells = ''.join('<td><a href="images/%s"><img style="width:%dpx" src="images/%s"></a><br><p>%s</p></td>'
                        % (n, width, n, t) for n, t in zip(names, labels))
        self.rows.append('<h3>%s</h3>\n<table border="1" style="table-layout: fixed;"><tr>%s</tr></table>' % (header, cells))

    def save(self):
        with open(os.path.join(self.web_dir, 'index.html'), 'w') as f:
            f.write('<!DOCTYPE html>\n<html><head><title>%s</title></head><body>\n%s\n</body></html>\n'
                    % (self.title, '\n'.join(self.rows)))


def resize_aspect(im, aspect_ratio):
    """widen (ratio > 1) or heighten (ratio < 1) with bicubic interpolation, as save_images does"""
    if aspect_ratio == 1.0:
        return im
    from PIL import Image
    h, w, _ = im.shape
    size = (int(w * aspect_ratio), h) if aspect_ratio > 1.0 else (w, int(h / aspect_ratio))
    return np.asarray(Image.fromarray(im).resize(size, Image.BICUBIC))


def save_images(page, visuals, image_path, aspect_ratio=1.0, width=256):
    stem = os.path.splitext(ntpath.basename(image_path[0]))[0]
    names, labels = [], []
    for label, tensor in visuals.items():
        im = resize_aspect(util.tensor2im(tensor), aspect_ratio)
        image_name = '%s_%s.png' % (stem, label)
        util.save_image(im, os.path.join(page.get_image_dir(), image_name))
        names.append(image_name)
        labels.append(label)
    page.add_row(stem, names, labels, width)
    return names


def main():
    opt = TestOptions().parse()
    opt.nThreads = 1            # the reference's test loop: one worker, one image, fixed order, no augmentation
    opt.batchSize = 1
    opt.serial_batches = True
    opt.no_flip = True
    opt.display_id = -1
    opt.sorted = True
    dataset = CreateDataLoader(opt).load_data()
    model = create_model(opt)
    model.save_networks('test')
    model.setup(opt)
    web_dir = os.path.join(opt.results_dir, opt.name, '%s_%s' % (opt.phase, opt.which_epoch))
    page = ResultPage(web_dir, 'Experiment = %s, Phase = %s, Epoch = %s' % (opt.name, opt.phase, opt.which_epoch))
    for i, data in enumerate(dataset):
        if i >= opt.how_many:
            break
        model.set_input(data)
        model.test()
        visuals = model.get_current_visuals()
        img_path = model.get_image_paths()
        if i % 5 == 0:
            print('processing (%04d)-th image... %s' % (i, img_path))
        save_images(page, visuals, img_path, aspect_ratio=opt.aspect_ratio, width=opt.display_winsize)
    page.save()


if __name__ == '__main__':
    main()
