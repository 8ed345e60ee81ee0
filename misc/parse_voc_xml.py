# coding: utf-8
"""PASCAL VOC annotations -> the annotation txt files every script here reads
(`index image_path width height [class x_min y_min x_max y_max]...` per line) - the reference's misc/parse_voc_xml.py with
its constants turned into options: objects marked difficult are skipped, images without a remaining object or without a
file on disk are skipped, indices count the lines written.

    python misc/parse_voc_xml.py --names ./voc_names.txt \\
        --train /data/VOCdevkit/VOC2007:trainval /data/VOCdevkit/VOC2012:trainval --val /data/VOCdevkit/VOC2007:test \\
        --train_out train.txt --val_out val.txt
"""
import argparse
import os
import sys
import xml.etree.ElementTree as ET


def read_names(path):
    with open(path) as f:
        return {line.strip(): i for i, line in enumerate(l for l in f if l.strip())}


def parse_xml(path, names):
    """[width, height, class, x_min, y_min, x_max, y_max, ...] (strings, as written to the file) or None when the image
    has no object that is not 'difficult'."""
    root = ET.parse(path)
    fields = [root.findtext('./size/width'), root.findtext('./size/height')]
    for obj in root.findall('object'):
        if (obj.findtext('difficult') or '0').strip() == '1':
            continue
        box = obj.find('bndbox')
        fields.append(str(names[obj.findtext('name').strip()]))
        fields.extend(box.findtext(tag).strip() for tag in ('xmin', 'ymin', 'xmax', 'ymax'))
    return fields if len(fields) > 2 else None


def write_split(out_path, sources, names, require_image=True):
    """sources: [(VOC year directory, image set name)], e.g. ('/data/VOCdevkit/VOC2007', 'trainval').  Returns the count."""
    count = 0
    with open(out_path, 'w') as out:
        for root, image_set in sources:
            with open(os.path.join(root, 'ImageSets', 'Main', image_set + '.txt')) as f:
                stems = [line.split()[0] for line in f if line.strip()]
            for stem in stems:
                fields = parse_xml(os.path.join(root, 'Annotations', stem + '.xml'), names)
                image = os.path.join(root, 'JPEGImages', stem + '.jpg')
                if fields is None or (require_image and not os.path.exists(image)):
                    continue
                out.write(' '.join([str(count), image] + fields) + '\n')
                count += 1
    return count


def _source(text):
    root, _, image_set = text.rpartition(':')
    if not root:
        raise argparse.ArgumentTypeError("expected <VOC year directory>:<image set>, got %r" % text)
    return root, image_set


def main(argv=None):
    ap = argparse.ArgumentParser(description="VOC xml annotations -> train / val annotation txt files")
    ap.add_argument('--names', default='./voc_names.txt', help="class names, one per line (the line number is the class id)")
    ap.add_argument('--train', nargs='*', type=_source, default=[], metavar='DIR:SET')
    ap.add_argument('--val', nargs='*', type=_source, default=[], metavar='DIR:SET')
    ap.add_argument('--train_out', default='train.txt')
    ap.add_argument('--val_out', default='val.txt')
    args = ap.parse_args(sys.argv[1:] if argv is None else argv)
    names = read_names(args.names)
    done = {}
    if args.train:
        done['train'] = write_split(args.train_out, args.train, names)
    if args.val:
        done['val'] = write_split(args.val_out, args.val, names)
    for split, n in done.items():
        print('%s: %d images' % (split, n))
    return done


if __name__ == '__main__':
    main()
